set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python bench.py --steps 10 --warmup 5 --batch 256 --no_cpu_baseline > gpurun_out/bench_b256.log 2>&1
tail -3 gpurun_out/bench_b256.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1a -o r1a -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1a.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof_r1a | head -20
find gpurun_out/prof_r1a -name "*kernel_stats*" | head
f=$(find gpurun_out/prof_r1a -name "*kernel_stats.csv" | head -1)
head -45 "$f"
# keep the output small: drop the big trace
find gpurun_out/prof_r1a -name "*kernel_trace.csv" -size +20M -delete
