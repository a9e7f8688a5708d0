# Round 6, GPU call 10: plain 1x1 products with K >= 256 on the staged implicit GEMM (the teacher's prologue-free conv3 of stage 3,
# backward-data of stage-3 conv1) -- kernel tests, A/B against the variant library with the old threshold (K >= 512)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_bench_geometry_gpu.py tests/test_igemm_gpu.py -m gpu -q --tb=short 2>&1 | tail -6 | cut -c1-300
run() {  # label, env...
  label=$1; shift
  v=$(env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); sg = d['config'].get('step_graph'); print('%.0f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], 'recorded' if sg else 'launch by launch'))
")
  echo "$label | $v" | tee -a $O/r06_plain_mink_ab.txt
  [ -z "$v" ] && tail -3 $O/r06_ab_err.txt
}
rm -f $O/r06_plain_mink_ab.txt
V=$GRAFT_REPO_ROOT/tools/gpu/_build/libpocketflow_hip_mink512.so
run "plain 1x1 on the staged kernel from K = 512 (variant library: rounds 2-5)" PF_HIP_LIB=$V
run "plain 1x1 on the staged kernel from K = 256 (product)                    " PF_X=0
run "plain 1x1 on the staged kernel from K = 512 (variant library: rounds 2-5)" PF_HIP_LIB=$V
run "plain 1x1 on the staged kernel from K = 256 (product)                    " PF_X=0
v=$(timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --step_graph 0 2>/dev/null | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('%.0f images/s  %.2f ms/step  launch by launch (host submit median %.1f ms)' % (d['value'], d['ms_per_step'], d['host_submit_ms_min_median_max'][1]))
")
echo "launch by launch (--step_graph 0), product | $v" | tee -a $O/r06_plain_mink_ab.txt
