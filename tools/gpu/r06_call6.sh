# Round 6, GPU call 6: the window-staged 3x3 kernel for 64 channels at 56 x 56 -- tests, per-layer timing, step A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_igemm_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q --tb=short -x -k "c64 or conv3x3 or statistics_and_residual or fwd_matches_torch or inference_bn or teacher_conv2" 2>&1 | tail -15 | cut -c1-400
timeout 300 python - <<'PY' 2>&1 | tail -8
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tools', 'gpu'))
from pocketflow_amd import hip
from _timing import gpu_time_us as timeit
B, H, C = 256, 56, 64
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(B, H, H, C, device='cuda', generator=g).bfloat16()
w = (torch.randn(C, 3, 3, C, device='cuda', generator=g) * 0.05).bfloat16()
y = torch.empty(B, H, H, C, device='cuda', dtype=torch.bfloat16)
M = B * H * H
bnx = torch.randn(M, C, device='cuda', generator=g).bfloat16()
ss = torch.stack([torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')]); mi = ss.clone()
print('3x3 64 -> 64 at 56 x 56, B = 256, us per launch (floors: MFMA 23.7, HBM 32.6)')
for on in ('0', '1'):
  os.environ['PF_CONV3X3_C64'] = on; hip.tuning_reload()
  G = hip.conv2d_stats_groups(M, C, geom=(B, H, H, C, C, 3, 3, 1, 1, 1, H, H))
  p = torch.empty(G, 4, C, device='cuda'); pb = torch.empty(G, 2, C, device='cuda')
  t0 = timeit(lambda: hip.conv2d_fwd(x, w, y, B, H, H, C, C, 3, 3, 1, 1, 1, H, H))
  t1 = timeit(lambda: hip.conv2d_fwd(x, w, y, B, H, H, C, C, 3, 3, 1, 1, 1, H, H, partial=p))
  t2 = timeit(lambda: hip.conv2d_fwd(x, w, y, B, H, H, C, C, 3, 3, 1, 1, 1, H, H, partial=pb, bn_x=bnx, bn_scale_shift=ss, bn_mean_invstd=mi, bn_act='Relu'))
  t3 = timeit(lambda: hip.conv2d_fwd(x, w, y, B, H, H, C, C, 3, 3, 1, 1, 1, H, H, out_scale_shift=ss, out_act='Relu'))
  print('%s | plain %6.1f | statistics %6.1f | backward-data + BN sums %6.1f | output affine %6.1f' % ('window kernel ' if on == '1' else 'per-tap kernel', t0, t1, t2, t3))
PY
run() {  # label, env...
  label=$1; shift
  v=$(env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); sg = d['config'].get('step_graph'); print('%.0f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], 'recorded' if sg else 'launch by launch'))
")
  echo "$label | $v" | tee -a $O/r06_conv3x3_c64_ab.txt
  [ -z "$v" ] && tail -3 $O/r06_ab_err.txt
}
rm -f $O/r06_conv3x3_c64_ab.txt
run "per-tap kernel (PF_CONV3X3_C64=0)" PF_CONV3X3_C64=0
run "window kernel (default)          " PF_X=0
run "per-tap kernel (PF_CONV3X3_C64=0)" PF_CONV3X3_C64=0
run "window kernel (default)          " PF_X=0
