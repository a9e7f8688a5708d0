cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 --tb=line -x -k "not fused_path" 2>&1 | tail -6 | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 5 --no_cpu_baseline 2>&1 | tail -1 | cut -c1-420
