set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
nproc; lscpu | grep "Model name"
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -30 > gpurun_out/pytest2.log
tail -30 gpurun_out/pytest2.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1b -o r1b -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 6 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1b.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/prof_r1b.log
find gpurun_out/prof_r1b -name '*kernel_trace.csv' -size +30M -delete
ls -la gpurun_out/prof_r1b/*
