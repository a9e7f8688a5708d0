# Round 6, GPU call 13: 256-input-channel tiles of k_wrw2 for the small-N 1x1 layers -- tests, per-layer table, step A/B; regression of
# the step-level tests after the MIOpen fallbacks of _Conv2dIgemm.backward moved to k_convg
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_bench_geometry_gpu.py tests/test_igemm_gpu.py -m gpu -q --tb=short 2>&1 | tail -6 | cut -c1-300
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_learner_gpu.py -m gpu -q --tb=short -x -k "not benchmarked_geometry" 2>&1 | tail -6 | cut -c1-300
for on in 0 1; do
  echo "PF_WRW2_TK256=$on" | tee -a $O/r06_wrw_tk256.txt
  PF_WRW2_TK256=$on timeout 600 python tools/gpu/wrw_layers.py 2>/dev/null | grep -E "conv1|per step" | tee -a $O/r06_wrw_tk256.txt | cut -c1-120
done
run() {  # label, env...
  label=$1; shift
  v=$(env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); sg = d['config'].get('step_graph'); print('%.0f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], 'recorded' if sg else 'launch by launch'))
")
  echo "$label | $v" | tee -a $O/r06_wrw_tk256.txt
  [ -z "$v" ] && tail -3 $O/r06_ab_err.txt
}
run "128-channel tiles (PF_WRW2_TK256=0)" PF_WRW2_TK256=0
run "256-channel tiles for N <= 128      " PF_X=0
run "128-channel tiles (PF_WRW2_TK256=0)" PF_WRW2_TK256=0
run "256-channel tiles for N <= 128      " PF_X=0
