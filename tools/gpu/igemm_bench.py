"""pf_conv2d_fwd (implicit GEMM, pf_igemm.hip) vs MIOpen on the ResNet-50 B=256 3x3 shapes (forward, backward-data) and the
plain 1x1 GEMM shapes of stages 3-4; TFLOP/s = 2*M*N*K_total / time.  PF_IGEMM_TILE is swept per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from pocketflow_amd import hip
for k in ('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
  os.environ.setdefault(k, '0')
torch.backends.cudnn.benchmark = True


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import gpu_time_us as timeit   # hipGraph replay: no host launch overhead in the numbers


B = int(os.environ.get('B', 256))
# (H, C, N, k, stride)
shapes = [(56, 64, 64, 3, 1), (56, 128, 128, 3, 2), (28, 128, 128, 3, 1), (28, 256, 256, 3, 2), (14, 256, 256, 3, 1),
          (14, 512, 512, 3, 2), (7, 512, 512, 3, 1), (14, 1024, 256, 1, 1), (14, 256, 1024, 1, 1), (7, 2048, 512, 1, 1),
          (7, 512, 2048, 1, 1), (28, 512, 128, 1, 1)]
tiles = ['256x128', '128x128', '256x64', '128x64']
print('%-22s | %-44s | dispatched | miopen fwd | miopen bwd-data | igemm best TF' % ('H,C,N,k,s', 'igemm us by tile ' + ' '.join(tiles)))
for H, C, N, k, s in shapes:
  g = torch.Generator(device='cuda').manual_seed(H + C + N)
  x = torch.randn(B, H, H, C, device='cuda', generator=g).bfloat16()
  w = (torch.randn(N, k, k, C, device='cuda', generator=g) * 0.05).bfloat16()
  pad = (k - 1) // 2
  Ho = (H + 2 * pad - k) // s + 1
  y = torch.empty(B, Ho, Ho, N, device='cuda', dtype=torch.bfloat16)
  M = B * Ho * Ho
  G = 0
  ts = []
  for t in tiles:
    if N % int(t.split('x')[1]):
      ts.append(float('nan')); continue
    os.environ['PF_IGEMM_TILE'] = t
    hip.tuning_reload()          # the library reads its switches once
    G = hip.conv2d_stats_groups(M, N, geom=(B, H, H, C, N, k, k, s, pad, pad, Ho, Ho))
    partial = torch.empty(G, 4, N, device='cuda')
    ts.append(timeit(lambda: hip.conv2d_fwd(x, w, y, B, H, H, C, N, k, k, s, pad, pad, Ho, Ho, partial=partial)))
  os.environ.pop('PF_IGEMM_TILE', None)
  hip.tuning_reload()          # the library reads its switches once
  # what the step runs: the dispatcher's own choice (for 3x3, 64 -> 64 at 56 x 56 the window-staged kernel of pf_conv3x3_c64.hip)
  G = hip.conv2d_stats_groups(M, N, geom=(B, H, H, C, N, k, k, s, pad, pad, Ho, Ho))
  partial = torch.empty(G, 4, N, device='cuda')
  t_def = timeit(lambda: hip.conv2d_fwd(x, w, y, B, H, H, C, N, k, k, s, pad, pad, Ho, Ho, partial=partial))
  x4 = x.permute(0, 3, 1, 2)
  w4 = w.permute(0, 3, 1, 2)
  ref = F.conv2d(x4, w4, stride=s, padding=pad)
  err = float(((y.permute(0, 3, 1, 2).float() - ref.float()).abs() > (ref.float().abs() * 2 ** -6 + 5e-2)).float().mean())
  t_mi = timeit(lambda: F.conv2d(x4, w4, stride=s, padding=pad))
  dy4 = torch.randn_like(ref)
  t_bd = timeit(lambda: torch.ops.aten.convolution_backward(dy4, x4, w4, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]))
  t_wr = timeit(lambda: torch.ops.aten.convolution_backward(dy4, x4, w4, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))
  fl = 2.0 * M * N * C * k * k
  best = min(t for t in ts if t == t)
  best = min(best, t_def)
  print('%-22s | %s (err %.0e) | %7.0f | %7.0f (%4.0f TF) | %7.0f | %4.0f TF | miopen wrw %7.0f' % (
      '%d,%d,%d,%d,%d' % (H, C, N, k, s), ' '.join('%7.0f' % t for t in ts), err, t_def, t_mi, fl / t_mi * 1e-6, t_bd, fl / best * 1e-6, t_wr))
