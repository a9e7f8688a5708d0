"""The roofline region of bench.py layer by layer: every 1x1 forward convolution of the ResNet-50 (v2 bottleneck) step at
B = 256 in the configuration the product launches it (producer BN + ReLU prologue, fake-quant for the student, statistics for
the consumer BN, shortcut in the epilogue of conv3), student and teacher, against the HBM floor of its algorithmic bytes.
The weighted sums are what `roofline.avg_launch_ms` of bench.py averages."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import gpu_time_us as timeit

B = int(os.environ.get('B', 256))
# (name, input H, K, N, stride, residual, statistics, launches per network)
LAYERS = [('s1 conv1 (pool out)', 56, 64, 64, 1, 0, 1, 1), ('s1 conv1', 56, 256, 64, 1, 0, 1, 2), ('s1 conv3', 56, 64, 256, 1, 1, 1, 3),
          ('s1 proj', 56, 64, 256, 1, 0, 0, 1), ('s2 conv1 @56', 56, 256, 128, 1, 0, 1, 1), ('s2 conv1', 28, 512, 128, 1, 0, 1, 3),
          ('s2 conv3', 28, 128, 512, 1, 1, 1, 4), ('s2 proj /2', 56, 256, 512, 2, 0, 0, 1), ('s3 conv1 @28', 28, 512, 256, 1, 0, 1, 1),
          ('s3 conv1', 14, 1024, 256, 1, 0, 1, 5), ('s3 conv3', 14, 256, 1024, 1, 1, 1, 6), ('s3 proj /2', 28, 512, 1024, 2, 0, 0, 1),
          ('s4 conv1 @14', 14, 1024, 512, 1, 0, 1, 1), ('s4 conv1', 7, 2048, 512, 1, 0, 1, 2), ('s4 conv3', 7, 512, 2048, 1, 1, 1, 3),
          ('s4 proj /2', 14, 1024, 2048, 2, 0, 0, 1)]
print('%-20s %-18s n | student us  teacher us | floor us (6.3 TB/s) | MFMA us (2.5 PF) | x floor' % ('layer', 'H,K,N,stride'))
tot = {'s': 0.0, 't': 0.0, 'f': 0.0, 'bytes': 0.0}
for name, H, K, N, stride, res, stats, cnt in LAYERS:
  Ho = H // stride
  M = B * Ho * Ho
  g = torch.Generator(device='cuda').manual_seed(H + K + N)
  X = torch.randn(B * H * H, K, device='cuda', generator=g).bfloat16()
  W = (torch.randn(N, K, device='cuda', generator=g) * 0.05).bfloat16()
  R = torch.randn(M, N, device='cuda', generator=g).bfloat16() if res else None
  ss = torch.stack([torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g)])
  slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
  hip.minmax_tensor(torch.relu(X.float() * ss[0] + ss[1]).contiguous(), slot)
  Y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  geom = (Ho, Ho, H, H, stride) if stride != 1 else None
  partial = None
  if stats:
    G = hip.conv1x1_stats_groups(M, N, K, prologue=True)
    partial = torch.empty(G, 4, N, device='cuda')
  ts = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, R=R, scale_shift=ss, act='Relu', slot=slot, bits=8, partial=partial, geom=geom))
  tt = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, R=R, scale_shift=ss, act='Relu', partial=partial, geom=geom))
  nbytes = (M * K + (2 if res else 1) * M * N) * 2
  floor = nbytes / 6.3e12 * 1e6
  mfma = 2.0 * M * N * K / 2.5e15 * 1e6
  print('%-20s %-18s %d | %10.0f  %10.0f | %8.0f | %8.0f | %.2f' % (name, '%d,%d,%d,%d' % (H, K, N, stride), cnt, ts, tt, floor, mfma,
                                                                    ts / max(floor, mfma)))
  tot['s'] += cnt * ts; tot['t'] += cnt * tt; tot['f'] += 2 * cnt * max(floor, mfma); tot['bytes'] += 2 * cnt * nbytes
n = 2 * sum(l[-1] for l in LAYERS)
print('per step: student %.2f ms + teacher %.2f ms = %.2f ms over %d launches; floor %.2f ms; %.1f MB/launch -> %.0f GB/s'
      % (tot['s'] / 1e3, tot['t'] / 1e3, (tot['s'] + tot['t']) / 1e3, n, tot['f'] / 1e3, tot['bytes'] / n / 1e6,
         tot['bytes'] / ((tot['s'] + tot['t']) * 1e-6) / 1e9))
