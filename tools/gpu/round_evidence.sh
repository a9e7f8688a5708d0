# Round evidence on the GPU box:  bash tools/gpu/round_evidence.sh [tag]   (tag defaults to r02; outputs gpurun_out/<tag>_*)
# smoke -> full GPU test suite -> rocprofv3 kernel trace of bench.py (stats + steady-state step table) -> separate PMC
# passes (FETCH_SIZE / WRITE_SIZE; never combined with a trace domain) -> the default bench.py line with cpu_baseline
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 --tb=line 2>&1 | tail -8 | cut -c1-300 > gpurun_out/${TAG}_pytest_gpu.log
cat gpurun_out/${TAG}_pytest_gpu.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/${TAG}_prof.log | cut -c1-200
python tools/prof_summary.py $(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/${TAG}_step_kernels_b256.csv | head -14 | cut -c1-150
cp $(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_rocprofv3_stats_b256.csv
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $c --out $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$c.csv | head -8 | cut -c1-160
done
cd $GRAFT_REPO_ROOT
PF_BENCH_TRACE_STEPS=1 timeout 900 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; grep -v '"metric"' gpurun_out/${TAG}_bench.log | tail -2 | cut -c1-300; grep '"metric"' gpurun_out/${TAG}_bench.log | cut -c1-2200
