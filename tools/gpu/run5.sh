cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout=600 --tb=line 2>&1 | tail -30 > gpurun_out/pytest5.log
cat gpurun_out/pytest5.log
