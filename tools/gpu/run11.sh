cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --timeout=300 --tb=line -s -k "fused_path" 2>&1 | grep -v Warning | tail -12 | cut -c1-900
