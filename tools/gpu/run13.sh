cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 --tb=line 2>&1 | tail -12 | cut -c1-300 > gpurun_out/pytest13.log
cat gpurun_out/pytest13.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r1e -o r1e -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1e.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/prof_r1e.log | cut -c1-300
python tools/prof_summary.py $(find /tmp/prof_r1e -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r01_step_kernels_fused2.csv | head -24 | cut -c1-160
timeout 600 python bench.py --steps 10 --warmup 5 2>&1 | tail -1 > gpurun_out/bench13.log; cut -c1-1500 gpurun_out/bench13.log
