"""Backward-filter (kernel + cross-split reduction, through the public entry points) per ResNet-50 layer at B = 256 as a function
of the number of workgroups a launch aims at (PF_WRW2_TARGET -> pixel splits -> fp32 slab traffic vs parallelism)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import gpu_time_us as timeit

B = 256
# (H, C, N, k, launches per step)
LAYERS = [(56, 64, 64, 1, 1), (56, 256, 64, 1, 2), (56, 64, 256, 1, 4), (56, 256, 128, 1, 1), (28, 512, 128, 1, 3), (28, 128, 512, 1, 4),
          (28, 512, 256, 1, 1), (14, 1024, 256, 1, 5), (14, 256, 1024, 1, 6), (14, 1024, 512, 1, 1), (7, 2048, 512, 1, 2), (7, 512, 2048, 1, 3),
          (28, 128, 128, 3, 3), (14, 256, 256, 3, 5), (7, 512, 512, 3, 2)]
TARGETS = [int(t) for t in os.environ.get('TARGETS', '256,384,512,768,1024').split(',')]
tot = {t: 0.0 for t in TARGETS}
print('%-16s n |' % 'H,C,N,k' + ''.join('   %5d (S)  ' % t for t in TARGETS))
for H, C, N, k, n in LAYERS:
  g = torch.Generator(device='cuda').manual_seed(H + C + N)
  M = B * H * H
  x = torch.randn(B, H, H, C, device='cuda', generator=g).bfloat16()
  dy = (torch.randn(B, H, H, N, device='cuda', generator=g) * 0.1).bfloat16()
  dw = torch.empty(N, k, k, C, device='cuda', dtype=torch.bfloat16)
  line = '%-16s %d |' % ('%d,%d,%d,%d' % (H, C, N, k), n)
  for t in TARGETS:
    os.environ['PF_WRW2_TARGET'] = str(t)
    hip.tuning_reload()          # the library reads its switches once
    if k == 1:
      S = hip.conv1x1_wrw_splits(M, N, C)
      ws = torch.empty((S + 32) * N * C, device='cuda')
      us = timeit(lambda: hip.conv1x1_wrw(dy, x, dw, ws, M, N, C))
    else:
      S = hip.conv2d_wrw_splits(M, N, C, k * k)
      ws = torch.empty((S + 32) * N * k * k * C, device='cuda')
      us = timeit(lambda: hip.conv2d_wrw(dy, x, dw, ws, B, H, H, C, N, k, k, 1, 1, 1, H, H))
    tot[t] += n * us
    line += '  %6.0f (%3d)' % (us, S)
  print(line)
print('per step (ms):     |' + ''.join('  %6.3f      ' % (tot[t] / 1e3) for t in TARGETS))
