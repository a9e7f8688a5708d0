cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 --tb=line 2>&1 | tail -8 | cut -c1-300 > gpurun_out/pytest23.log
cat gpurun_out/pytest23.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r1f -o r1f -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1f.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/prof_r1f.log | cut -c1-200
python tools/prof_summary.py $(find /tmp/prof_r1f -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r01_step_kernels_final.csv | head -16 | cut -c1-150
cp $(find /tmp/prof_r1f -name '*kernel_stats.csv' | head -1) gpurun_out/r01_kernel_stats_final.csv
timeout 900 python bench.py > gpurun_out/bench23.log 2>&1; tail -1 gpurun_out/bench23.log | cut -c1-1800
