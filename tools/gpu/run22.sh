cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --timeout=300 --tb=line -k "wrw" 2>&1 | tail -3 | cut -c1-300
NO_MIOPEN=1 timeout 300 python tools/gpu/conv_bench.py 2>&1 | tail -12 | cut -c1-20,85-
