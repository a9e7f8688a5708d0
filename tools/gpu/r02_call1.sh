# round 2, GPU call 1: hardware probe, stream-kernel correctness + A/B, BASELINE-config parity tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/gpu/probe_lds.hip -o /tmp/probe_lds 2>/dev/null && timeout 60 /tmp/probe_lds > gpurun_out/r02_probe_lds.log 2>&1; tail -3 gpurun_out/r02_probe_lds.log
timeout 900 python -m pytest tests/test_conv_gpu.py -q --tb=short -x -k "not as_accurate" 2>&1 | tail -15 | cut -c1-250 > gpurun_out/r02_c1_conv_tests.log; cat gpurun_out/r02_c1_conv_tests.log
timeout 600 python tools/gpu/conv_bench2.py > gpurun_out/r02_c1_conv_bench2.log 2>&1; tail -16 gpurun_out/r02_c1_conv_bench2.log | cut -c1-200
timeout 900 python -m pytest tests/test_parity_gpu.py -q --tb=short -k "resnet50 or cp_mobilenet" 2>&1 | tail -25 | cut -c1-300 > gpurun_out/r02_c1_parity.log; cat gpurun_out/r02_c1_parity.log
timeout 600 python -m pytest tests/test_conv_gpu.py -q --tb=short -k "as_accurate" 2>&1 | tail -8 | cut -c1-400 > gpurun_out/r02_c1_fused_acc.log; cat gpurun_out/r02_c1_fused_acc.log
