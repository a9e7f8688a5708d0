cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 --tb=line 2>&1 | tail -5 | cut -c1-300
