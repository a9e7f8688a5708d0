# Round 6, GPU call 14: (a) 8-bit activation codes as the igemm input operand (variant library), (b) k_bn_finalize with 16 channels x
# 64 lanes per workgroup -- BN tests, step A/B against the previous library kept as a variant is not possible in one tree, so:
# kernel time from a short rocprofv3 trace of bench.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python tools/gpu/a8_codes_probe.py 2>&1 | grep -v Warning | tee $O/r06_a8_codes_ab.txt | cut -c1-250
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_igemm_gpu.py tests/test_conv_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q --tb=short 2>&1 | tail -5 | cut -c1-300
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -x -k "not benchmarked_geometry" 2>&1 | tail -4 | cut -c1-300
for i in 1 2; do timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('%.0f images/s  %.2f ms/step' % (d['value'], d['ms_per_step']))
"; done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof14 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no_cpu_baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof14/*/*kernel_stats.csv $O/prof14/*kernel_stats.csv 2>/dev/null | head -1)
grep -E "k_bn_finalize|k_bn_bwd_finalize|k_wrw_reduce" $f | cut -c1-200
exit 0
