cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_learner_gpu.py -m gpu -q --timeout=600 --tb=short -k "two_ranks" 2>&1 | tail -25 | cut -c1-400
