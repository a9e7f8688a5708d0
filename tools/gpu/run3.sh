set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x --timeout=600 2>&1 | tail -40 > gpurun_out/pytest3.log
tail -40 gpurun_out/pytest3.log
timeout 300 python tools/gpu/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1; cat gpurun_out/gemm_bench.log
cd /tmp
( time timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r1c -o r1c -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 6 --batch 256 --no_cpu_baseline ) > $GRAFT_REPO_ROOT/gpurun_out/prof_r1c.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/prof_r1c.log; tail -4 gpurun_out/prof_r1c.log
python tools/prof_summary.py $(find /tmp/prof_r1c -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r01_step_kernels.csv | head -70
