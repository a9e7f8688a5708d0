cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_conv_gpu.py -m gpu -q --timeout=600 --tb=line 2>&1 | grep -v "^  \|Warning\|^$" | tail -14 | cut -c1-500
timeout 600 python bench.py --steps 10 --warmup 5 --no_cpu_baseline 2>&1 | tail -1 | cut -c1-420
