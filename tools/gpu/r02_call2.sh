# round 2, GPU call 2: implicit-GEMM kernel correctness + bench vs MIOpen; stream kernel after the prefetch rework
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_igemm_gpu.py -q --tb=short 2>&1 | tail -25 | cut -c1-250 > gpurun_out/r02_c2_igemm_tests.log; cat gpurun_out/r02_c2_igemm_tests.log
timeout 900 python -m pytest tests/test_conv_gpu.py -q --tb=short -k "not as_accurate" 2>&1 | tail -25 | cut -c1-250 > gpurun_out/r02_c2_conv_tests.log; cat gpurun_out/r02_c2_conv_tests.log
SHAPES="56,64,64,0;56,64,256,1;56,256,64,0;56,256,128,0;28,128,512,1;28,256,128,0" timeout 600 python tools/gpu/conv_bench2.py > gpurun_out/r02_c2_conv_bench2.log 2>&1; tail -8 gpurun_out/r02_c2_conv_bench2.log | cut -c1-200
timeout 900 python tools/gpu/igemm_bench.py > gpurun_out/r02_c2_igemm_bench.log 2>&1; tail -14 gpurun_out/r02_c2_igemm_bench.log | cut -c1-220
timeout 600 python -m pytest tests/test_parity_gpu.py -q --tb=short -k "resnet50_distillation_matches_oracle and 64 or cp_mobilenet" 2>&1 | tail -12 | cut -c1-300 > gpurun_out/r02_c2_parity.log; cat gpurun_out/r02_c2_parity.log
