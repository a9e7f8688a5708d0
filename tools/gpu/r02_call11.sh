# round 2, GPU call 11: why is the FIRST bench process on a fresh box 5x slower than the second?  kernel table of both
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
ls -la ~/.config/miopen ~/.cache/miopen 2>&1 | head
for r in first second; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$r -o $r -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_c11_bench_$r.log 2>&1)
  tail -1 gpurun_out/r02_c11_bench_$r.log | cut -c1-200
  f=$(find /tmp/prof_$r -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/r02_c11_kernel_stats_$r.csv
  head -12 $f | cut -c1-200
  find ~/.config/miopen ~/.cache/miopen -type f 2>/dev/null | head -20
done
