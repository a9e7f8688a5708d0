# Round 6, GPU call 11: strided-projection backward-data on the staged kernel with its row scatter -- tests + A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q --tb=short -k "ymap or projection" 2>&1 | tail -6 | cut -c1-300
run() {  # label, env...
  label=$1; shift
  v=$(env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); sg = d['config'].get('step_graph'); print('%.0f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], 'recorded' if sg else 'launch by launch'))
")
  echo "$label | $v" | tee -a $O/r06_plain_mink_ab2.txt
  [ -z "$v" ] && tail -3 $O/r06_ab_err.txt
}
rm -f $O/r06_plain_mink_ab2.txt
V=$GRAFT_REPO_ROOT/tools/gpu/_build/libpocketflow_hip_mink512.so
run "variant library: plain / row-mapped 1x1 on the staged kernel from K = 512" PF_HIP_LIB=$V
run "product: from K = 256, row-mapped launches included                     " PF_X=0
run "variant library: plain / row-mapped 1x1 on the staged kernel from K = 512" PF_HIP_LIB=$V
run "product: from K = 256, row-mapped launches included                     " PF_X=0
