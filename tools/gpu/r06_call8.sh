# Round 6, GPU call 8: the window-staged backward-filter kernel (3x3, 64 -> 64, 56 x 56) -- tests, timing, step A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_igemm_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q --tb=short -k "wrw or conv3x3" 2>&1 | tail -15 | cut -c1-400
timeout 300 python - <<'PY' 2>&1 | tail -6
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tools', 'gpu'))
from pocketflow_amd import hip
from _timing import gpu_time_us as timeit
B, H, C = 256, 56, 64
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(B, H, H, C, device='cuda', generator=g).bfloat16()
dy = (torch.randn(B, H, H, C, device='cuda', generator=g) * 0.1).bfloat16()
M = B * H * H
print('3x3 backward-filter 64 -> 64 at 56 x 56, B = 256, us per call incl. the slab reduction (floors: MFMA 23.7, HBM 32.6; MIOpen 165)')
for on in ('0', '1'):
  os.environ['PF_CONV3X3_C64'] = on; hip.tuning_reload()
  S = hip.conv2d_wrw_splits(M, C, C, 9)
  ws = torch.empty((S + 32) * C * 9 * C, device='cuda')
  dw = torch.empty(C, 3, 3, C, device='cuda')
  t = timeit(lambda: hip.conv2d_wrw(dy, x, dw, ws, B, H, H, C, C, 3, 3, 1, 1, 1, H, H))
  print('%s | %6.1f us (workspace for %d slabs)' % ('window kernel     ' if on == '1' else 'shared-tile kernel', t, S))
PY
run() {  # label, env...
  label=$1; shift
  v=$(env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); sg = d['config'].get('step_graph'); print('%.0f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], 'recorded' if sg else 'launch by launch'))
")
  echo "$label | $v" | tee -a $O/r06_wrw3x3_c64_ab.txt
  [ -z "$v" ] && tail -3 $O/r06_ab_err.txt
}
rm -f $O/r06_wrw3x3_c64_ab.txt
run "per-tap forward / shared-tile backward-filter (PF_CONV3X3_C64=0)" PF_CONV3X3_C64=0
run "window kernels, forward + backward-data + backward-filter     " PF_X=0
run "per-tap forward / shared-tile backward-filter (PF_CONV3X3_C64=0)" PF_CONV3X3_C64=0
run "window kernels, forward + backward-data + backward-filter     " PF_X=0
