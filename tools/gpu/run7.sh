cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --timeout=300 --tb=short 2>&1 | tail -40 > gpurun_out/pytest7.log
cat gpurun_out/pytest7.log
timeout 600 python tools/gpu/conv_bench.py > gpurun_out/conv_bench.log 2>&1; tail -20 gpurun_out/conv_bench.log
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout=600 --tb=line 2>&1 | grep -v "^  \|^$" | tail -30 > gpurun_out/pytest7b.log
cat gpurun_out/pytest7b.log
