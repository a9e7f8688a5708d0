"""Backward-filter, every shape of the ResNet-v2-50 step at batch B (default 256): ours (the dispatcher's own choice, quantising
prologue on the 1x1 layers as in the step) vs the HBM floor (operands read once at 6.3 TB/s) vs the MFMA floor (2.5 PFLOP/s)
vs MIOpen on materialised operands (aten.convolution_backward, benchmark mode).   python tools/gpu/wrw_layers.py > profiles/rNN_wrw_layers.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip
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
# (name, H of the input, K = C_in, N = C_out, window, stride, launches per step)
LAYERS = [('s1 conv1 (pool out)', 56, 64, 64, 1, 1, 1), ('s1 conv1', 56, 256, 64, 1, 1, 2), ('s1 conv2', 56, 64, 64, 3, 1, 3),
          ('s1 conv3', 56, 64, 256, 1, 1, 3), ('s1 proj', 56, 64, 256, 1, 1, 1),
          ('s2 conv1 @56', 56, 256, 128, 1, 1, 1), ('s2 conv2 /2', 56, 128, 128, 3, 2, 1), ('s2 conv1', 28, 512, 128, 1, 1, 3),
          ('s2 conv2', 28, 128, 128, 3, 1, 3), ('s2 conv3', 28, 128, 512, 1, 1, 4), ('s2 proj /2', 56, 256, 512, 1, 2, 1),
          ('s3 conv1 @28', 28, 512, 256, 1, 1, 1), ('s3 conv2 /2', 28, 256, 256, 3, 2, 1), ('s3 conv1', 14, 1024, 256, 1, 1, 5),
          ('s3 conv2', 14, 256, 256, 3, 1, 5), ('s3 conv3', 14, 256, 1024, 1, 1, 6), ('s3 proj /2', 28, 512, 1024, 1, 2, 1),
          ('s4 conv1 @14', 14, 1024, 512, 1, 1, 1), ('s4 conv2 /2', 14, 512, 512, 3, 2, 1), ('s4 conv1', 7, 2048, 512, 1, 1, 2),
          ('s4 conv2', 7, 512, 512, 3, 1, 2), ('s4 conv3', 7, 512, 2048, 1, 1, 3), ('s4 proj /2', 14, 1024, 2048, 1, 2, 1)]
print('# backward-filter per layer, B = %d, us; floors: HBM = (X + dY) read once at 6.3 TB/s, MFMA = 2 M N K taps / 2.5 PFLOP/s' % B)
print('%-20s %-18s %2s | %7s %7s | %6s %6s | %7s | %s' % ('layer', 'H,K,N,k,stride', 'n', 'ours', 'MIOpen', 'HBM', 'MFMA', 'x floor', 'splits'))
tot = tot_floor = tot_mi = 0.0
n_launch = 0
for name, H, K, N, k, s, cnt in LAYERS:
  pad = (k - 1) // 2
  Ho = (H + 2 * pad - k) // s + 1
  M = B * Ho * Ho
  g = torch.Generator(device='cuda').manual_seed(H + K + N + k)
  x = torch.randn(B, H, H, K, device='cuda', generator=g).bfloat16()
  dy = (torch.randn(B, Ho, Ho, N, device='cuda', generator=g) * 0.1).bfloat16()
  if k == 1:
    ss = torch.stack([torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g)])
    slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
    hip.minmax_tensor(torch.relu(x.float().reshape(-1, K) * ss[0] + ss[1]).contiguous(), slot)
    S = hip.conv1x1_wrw_splits(M, N, K)
    ws = torch.empty((S + 32) * N * K, device='cuda')
    dw = torch.empty(N, K, device='cuda')
    geom = (Ho, Ho, H, H, s) if s > 1 else None
    t = timeit(lambda: hip.conv1x1_wrw(dy, x, dw, ws, M, N, K, scale_shift=ss, act='Relu', slot=slot, bits=8, geom=geom))
  else:
    S = hip.conv2d_wrw_splits(M, N, K, k * k)
    ws = torch.empty((S + 32) * N * k * k * K, device='cuda')
    dw = torch.empty(N, k, k, K, device='cuda')
    t = timeit(lambda: hip.conv2d_wrw(dy, x, dw, ws, B, H, H, K, N, k, k, s, pad, pad, Ho, Ho))
  x4, dy4 = x.permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2)
  w4 = torch.zeros(N, K, k, k, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
  try:
    t_mi = timeit(lambda: torch.ops.aten.convolution_backward(dy4, x4, w4, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))
  except Exception:            # pylint: disable=broad-except
    t_mi = float('nan')
  rows_in = B * H * H if s == 1 else B * H * H          # a strided 1x1 touches every second row / column: whole 128-byte lines are still fetched
  hbm = (rows_in * K + M * N) * 2 / 6.3e12 * 1e6
  mfma = 2.0 * M * N * K * k * k / 2.5e15 * 1e6
  fl = max(hbm, mfma)
  print('%-20s %-18s %2d | %7.1f %7.1f | %6.1f %6.1f | %7.2f | %d' % (name, '%d,%d,%d,%d,%d' % (H, K, N, k, s), cnt, t, t_mi, hbm, mfma, t / fl, S))
  tot += t * cnt; tot_floor += fl * cnt; tot_mi += t_mi * cnt; n_launch += cnt
  del x, dy, dw, ws
print('per step: ours %.2f ms over %d launches (+ stem, dense); floors %.2f ms; MIOpen %.2f ms' % (tot / 1e3, n_launch, tot_floor / 1e3, tot_mi / 1e3))
