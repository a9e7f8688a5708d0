"""Backward-filter A/B on the ResNet-50 B=256 shapes: transposed-LDS-read kernel (pf_wrw.hip) vs the 2-byte-scatter kernel
(pf_conv.hip) for the 1x1 convolutions (with the quantising prologue), vs MIOpen for the 3x3 ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip
for k in ('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
  os.environ.setdefault(k, '0')
torch.backends.cudnn.benchmark = True


def timeit(fn, n=10):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n * 1e3


B = int(os.environ.get('B', 256))
print('1x1 (prologue, 8-bit):  HW,K,N | shared-tile us | scatter us | floor us (6.3 TB/s) | max rel err')
for hw, K, N in [(56, 64, 64), (56, 64, 256), (56, 256, 64), (56, 256, 128), (28, 128, 512), (28, 512, 128), (28, 512, 256),
                 (14, 256, 1024), (14, 1024, 256), (14, 1024, 512), (7, 512, 2048), (7, 2048, 512)]:
  M = B * hw * hw
  g = torch.Generator(device='cuda').manual_seed(hw + K + N)
  X = torch.randn(M, K, device='cuda', generator=g).bfloat16()
  dY = (torch.randn(M, N, device='cuda', generator=g) * 0.1).bfloat16()
  ss = torch.stack([torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g)])
  slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
  hip.minmax_tensor(torch.relu(X.float() * ss[0] + ss[1]).contiguous(), slot)
  res = {}
  for mode in ('1', '0'):
    os.environ['PF_WRW2'] = mode
    hip.tuning_reload()          # the library reads its switches once
    ws = torch.empty((hip.conv1x1_wrw_splits(M, N, K) + 32) * N * K, device='cuda')
    dW = torch.empty(N, K, device='cuda', dtype=torch.bfloat16)
    t = timeit(lambda: hip.conv1x1_wrw(dY, X, dW, ws, M, N, K, scale_shift=ss, act='Relu', slot=slot, bits=8))
    res[mode] = (t, dW.float().clone())
  os.environ.pop('PF_WRW2')
  hip.tuning_reload()          # the library reads its switches once
  err = float((res['1'][1] - res['0'][1]).abs().max() / (res['0'][1].abs().max() + 1e-9))
  print('%-14s | %7.0f | %7.0f | %6.0f | %.1e' % ('%d,%d,%d' % (hw, K, N), res['1'][0], res['0'][0], M * (K + N) * 2 / 6.3e12 * 1e6, err))
print('3x3:  H,C,N,stride | tr us | miopen us | TF tr')
for H, C, N, s in [(56, 64, 64, 1), (56, 128, 128, 2), (28, 128, 128, 1), (28, 256, 256, 2), (14, 256, 256, 1), (14, 512, 512, 2), (7, 512, 512, 1)]:
  g = torch.Generator(device='cuda').manual_seed(H + C)
  x = torch.randn(B, H, H, C, device='cuda', generator=g).bfloat16()
  Ho = (H + 2 - 3) // s + 1
  dy = (torch.randn(B, Ho, Ho, N, device='cuda', generator=g) * 0.1).bfloat16()
  M = B * Ho * Ho
  S = hip.conv2d_wrw_splits(M, N, C, 9)
  ws = torch.empty((S + 32) * N * 9 * C, device='cuda')
  dw = torch.empty(N, 3, 3, C, device='cuda', dtype=torch.bfloat16)
  t = timeit(lambda: hip.conv2d_wrw(dy, x, dw, ws, B, H, H, C, N, 3, 3, s, 1, 1, Ho, Ho))
  x4, dy4 = x.permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2)
  w4 = torch.zeros(N, C, 3, 3, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
  t_mi = timeit(lambda: torch.ops.aten.convolution_backward(dy4, x4, w4, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
  print('%-16s | %7.0f | %7.0f | %5.0f (S=%d)' % ('%d,%d,%d,%d' % (H, C, N, s), t, t_mi, 2.0 * M * N * C * 9 / t * 1e-6, S))
