set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout=600 2>&1 | tail -60 > gpurun_out/pytest4.log
tail -60 gpurun_out/pytest4.log
export MIOPEN_FIND_MODE=2
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  head -2 $f
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $c --out $GRAFT_REPO_ROOT/gpurun_out/r01_pmc_$c.csv
done
