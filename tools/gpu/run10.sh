cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_learner_gpu.py -m gpu -q --timeout=300 --tb=line -s -k "fused_path or channel_pruned or conv1x1" 2>&1 | grep -v Warning | tail -30 > gpurun_out/pytest10.log
cat gpurun_out/pytest10.log | cut -c1-600
timeout 600 python bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline 2>&1 | tail -1 | cut -c1-300
