cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --timeout=300 --tb=line -k "conv1x1" 2>&1 | tail -5 | cut -c1-300
NO_MIOPEN=1 timeout 300 python tools/gpu/conv_bench.py 2>&1 | tail -13
