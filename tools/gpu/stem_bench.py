"""ResNet stem (7x7/2, 3 -> 64) at the bench shape: pf_conv_stem_fwd vs MIOpen.  python tools/gpu/stem_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from pocketflow_amd import hip
for k in ('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
  os.environ.setdefault(k, '0')
torch.backends.cudnn.benchmark = True
B, H = int(os.environ.get('B', 256)), int(os.environ.get('H', 224))
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(B, H, H, 3, device='cuda', generator=g).bfloat16()
w = (torch.randn(64, 7, 7, 3, device='cuda', generator=g) * 0.1).bfloat16()
y = torch.empty(B, H // 2, H // 2, 64, device='cuda', dtype=torch.bfloat16)
xn, wn = x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2)


def timeit(fn, it=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(it):
    fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / it * 1e3


t_own = timeit(lambda: hip.conv_stem_fwd(x, w, y, B, H, H))
t_mi = timeit(lambda: F.conv2d(xn, wn, stride=2, padding=3))
ref = F.conv2d(xn.float(), wn.float(), stride=2, padding=3).permute(0, 2, 3, 1)
err = float((y.float() - ref).abs().max() / ref.abs().max())
nbytes = (x.numel() + y.numel()) * 2
print('stem %dx%dx%d: own %.0f us (%.2f TB/s of in+out bytes, %.0f TFLOP/s useful)  MIOpen %.0f us  max rel err %.1e  floor(6.3 TB/s) %.0f us' % (
    B, H, H, t_own, nbytes / t_own / 1e6, 2 * B * (H // 2) ** 2 * 64 * 147 / t_own / 1e6, t_mi, err, nbytes / 6.3e6))

# backward-filter
dy = (torch.randn(B, H // 2, H // 2, 64, device='cuda', generator=g) * 0.1).bfloat16()
S = hip.conv_stem_wrw_slabs(B, H, H)
if S > 0:
  ws = torch.empty((S + 32) * 64 * 147, device='cuda')
  dw = torch.empty(64, 7, 7, 3, device='cuda', dtype=torch.bfloat16)
  dyn = dy.permute(0, 3, 1, 2)
  t_own = timeit(lambda: hip.conv_stem_wrw(dy, x, dw, ws, B, H, H))
  t_mi = timeit(lambda: torch.ops.aten.convolution_backward(dyn, xn, wn, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False]))
  ref = torch.ops.aten.convolution_backward(dyn.float(), xn.float(), wn.float(), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                            [False, True, False])[1].permute(0, 2, 3, 1)
  err = float((dw.float() - ref).abs().max() / ref.abs().max())
  print('stem wrw: own %.0f us (incl. slab reduction; %d slabs)  MIOpen %.0f us  max rel err %.1e  floor %.0f us' % (
      t_own, S, t_mi, err, (x.numel() + dy.numel()) * 2 / 6.3e6))
