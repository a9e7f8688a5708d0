# round 2, GPU call 8: backward-filter kernel after the division-free pixel advance; fresh per-step kernel table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_igemm_gpu.py -q --tb=short -k "wrw" 2>&1 | tail -5 | cut -c1-300
timeout 600 python tools/gpu/wrw_bench.py > gpurun_out/r02_c8_wrw_bench.log 2>&1; tail -9 gpurun_out/r02_c8_wrw_bench.log | cut -c1-200
timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c8_bench.log 2>&1; tail -1 gpurun_out/r02_c8_bench.log | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c8 -o c8 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_c8_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find /tmp/prof_c8 -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r02_c8_step_kernels.csv | head -48 | cut -c1-150
