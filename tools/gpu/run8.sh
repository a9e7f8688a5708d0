cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_parity_gpu.py -m gpu -q --timeout=300 --tb=short 2>&1 | tail -40 > gpurun_out/pytest8.log
cat gpurun_out/pytest8.log
NO_MIOPEN=1 timeout 300 python tools/gpu/conv_bench.py > gpurun_out/conv_bench2.log 2>&1; tail -14 gpurun_out/conv_bench2.log
timeout 600 python bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > gpurun_out/bench_fused.log 2>&1; tail -1 gpurun_out/bench_fused.log | cut -c1-400
