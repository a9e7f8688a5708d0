cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -x -k "one_step_at_224_with_8bit" 2>&1 | tail -30 | cut -c1-600
echo "--- one queue"
PF_WRW_SIDE=0 timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=line -x -k "one_step_at_224_with_8bit" 2>&1 | tail -5 | cut -c1-600
exit 0
