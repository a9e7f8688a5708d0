cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/gpu/debug_strides.py 2>&1 | grep -v Warning | tail -20
