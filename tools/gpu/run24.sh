cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 600 python bench.py ) > gpurun_out/bench24.log 2>&1; grep '"metric"' gpurun_out/bench24.log | cut -c1-2000; tail -4 gpurun_out/bench24.log
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout=300 --tb=line -k "lenet" 2>&1 | tail -3 | cut -c1-300
