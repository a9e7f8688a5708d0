# round 2, GPU call 7: shared-tile backward-filter kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_igemm_gpu.py tests/test_conv_gpu.py -q --tb=short -k "wrw" 2>&1 | tail -15 | cut -c1-300 > gpurun_out/r02_c7_tests.log; cat gpurun_out/r02_c7_tests.log
timeout 600 python tools/gpu/wrw_bench.py > gpurun_out/r02_c7_wrw_bench.log 2>&1; tail -24 gpurun_out/r02_c7_wrw_bench.log | cut -c1-200
timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c7_bench.log 2>&1; tail -1 gpurun_out/r02_c7_bench.log | cut -c1-400
PF_OWN_CONV2D_WRW=0 timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c7_bench_miopenwrw.log 2>&1; tail -1 gpurun_out/r02_c7_bench_miopenwrw.log | cut -c1-300
