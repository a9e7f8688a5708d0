# round 2, GPU call 12: per-step kernel table of the FIRST bench process on a fresh box (empty MIOpen user db)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_first -o first -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_c12_bench_first.log 2>&1)
grep '"metric"' gpurun_out/r02_c12_bench_first.log | cut -c1-200
python tools/prof_summary.py $(find /tmp/prof_first -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r02_c12_step_kernels_first.csv | head -40 | cut -c1-170
