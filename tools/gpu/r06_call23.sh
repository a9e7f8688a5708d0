# Round 6, GPU call 23: the kernel trace of the default bench command at the round's last commit (rocprofv3 --kernel-trace --stats;
# no counters in this pass)
TAG=r06
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/${TAG}_prof.log | cut -c1-200
python tools/prof_summary.py $(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/${TAG}_step_kernels_b256.csv | head -14 | cut -c1-150
cp $(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_rocprofv3_stats_b256.csv
exit 0
