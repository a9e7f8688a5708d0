# round 2, GPU call 6: igemm with buffer-load staging + prologue variant; deterministic NUQ codebook gradient; split-bucket bins
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_igemm_gpu.py tests/test_kernels_gpu.py tests/test_conv_gpu.py -q --tb=short -k "not as_accurate" 2>&1 | tail -15 | cut -c1-300 > gpurun_out/r02_c6_tests.log; cat gpurun_out/r02_c6_tests.log
timeout 900 python tools/gpu/igemm_bench.py > gpurun_out/r02_c6_igemm_bench.log 2>&1; tail -14 gpurun_out/r02_c6_igemm_bench.log | cut -c1-230
SHAPES="28,512,128,0;28,512,256,0;14,256,1024,1;14,1024,256,0;14,1024,512,0;7,512,2048,1;7,2048,512,0" timeout 600 python tools/gpu/conv_bench2.py > gpurun_out/r02_c6_conv_bench2.log 2>&1; tail -8 gpurun_out/r02_c6_conv_bench2.log | cut -c1-200
timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c6_bench.log 2>&1; tail -1 gpurun_out/r02_c6_bench.log | cut -c1-400
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_conv_gpu.py -q --tb=short -k "nuq or as_accurate" 2>&1 | tail -5 | cut -c1-300
