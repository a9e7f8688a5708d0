cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== BN 128 default"; NO_MIOPEN=1 timeout 300 python tools/gpu/conv_bench.py 2>&1 | tail -12 | cut -c1-60
echo "== BN 64 forced"; PF_CONV_BN=64 NO_MIOPEN=1 timeout 300 python tools/gpu/conv_bench.py 2>&1 | tail -12 | cut -c1-60
