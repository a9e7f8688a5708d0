"""k_bn_bwd_apply (2.9 ms of the ResNet-50 step, 49 launches) over the step's BN shapes at B = 256 against the HBM floor of 2 reads +
1 write (+ the shortcut gradient where the block has one).  The variant columns (rows per loop trip x rows per block the grid is sized
for) belong to an experimental build of round 3 (environment switches PF_BN_BWD_ROWS / PF_BN_BWD_RPB, not in the tree: every variant
was within 1 % of or slower than the shipped 2 / 2 -- profiles/r03_bn_bwd_bench.txt); with the shipped library all columns time the
same kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import gpu_time_us as timeit

B = 256
# (H, C, launches per step, with addend): bn1 of every block reads the shortcut gradient as addend
SHAPES = [(112, 64, 1, 0), (56, 64, 6, 0), (56, 256, 3, 1), (28, 128, 8, 0), (28, 512, 4, 1), (56, 128, 1, 0), (14, 256, 12, 0),
          (14, 1024, 6, 1), (28, 256, 1, 0), (7, 512, 6, 0), (7, 2048, 3, 1), (14, 512, 1, 0)]
variants = [(2, 2), (4, 2), (2, 4), (4, 4), (4, 8), (2, 8)]
tot = {v: 0.0 for v in variants}
floor_tot = 0.0
print('%-18s n |' % 'H,C,addend' + ''.join('  rows%d/rpb%d' % v for v in variants) + ' | floor us')
for H, C, n, add in SHAPES:
  rows = B * H * H
  g = torch.Generator(device='cuda').manual_seed(H + C)
  dq = torch.randn(rows, C, device='cuda', generator=g).bfloat16()
  x = torch.randn(rows, C, device='cuda', generator=g).bfloat16()
  ad = torch.randn(rows, C, device='cuda', generator=g).bfloat16() if add else None
  dx = torch.empty_like(x)
  ss = torch.stack([torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')])
  mi = torch.stack([torch.randn(C, device='cuda'), torch.rand(C, device='cuda') + 0.5])
  dg, db = torch.randn(C, device='cuda'), torch.randn(C, device='cuda')
  line = '%-18s %d |' % ('%d,%d,%d' % (H, C, add), n)
  for v in variants:
    os.environ['PF_BN_BWD_ROWS'], os.environ['PF_BN_BWD_RPB'] = str(v[0]), str(v[1])
    t = timeit(lambda: hip.bn_bwd_apply(dq, x, dx, rows, C, ss, mi, dg, db, 'Relu', addend=ad))
    tot[v] += n * t
    line += '  %10.1f' % t
  fl = rows * C * 2 * (4 if add else 3) / 6.3e12 * 1e6
  floor_tot += n * fl
  print(line + ' | %6.1f' % fl)
print('per step (ms):     |' + ''.join('  %10.3f' % (tot[v] / 1e3) for v in variants) + ' | %6.3f' % (floor_tot / 1e3))
