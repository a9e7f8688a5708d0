cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MIOPEN_FIND_MODE=2
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc2_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $c --out $GRAFT_REPO_ROOT/gpurun_out/r01_pmc_$c.csv | head -12
done
