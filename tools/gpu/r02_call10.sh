# round 2, GPU call 10: three-stage wrw2 ring; seg_transpose effect; two-rank bench on one GPU; headline bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -q --tb=short -k "wrw" 2>&1 | tail -4 | cut -c1-300
timeout 600 python tools/gpu/wrw_bench.py > gpurun_out/r02_c10_wrw_bench.log 2>&1; tail -30 gpurun_out/r02_c10_wrw_bench.log | cut -c1-160
timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c10_bench.log 2>&1; tail -1 gpurun_out/r02_c10_bench.log | cut -c1-400
PF_SEG_TRANSPOSE=0 timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c10_bench_notr.log 2>&1; tail -1 gpurun_out/r02_c10_bench_notr.log | cut -c1-200
timeout 900 python -m pytest tests -q --tb=short -m gpu -k "two_ranks" 2>&1 | tail -4 | cut -c1-300
