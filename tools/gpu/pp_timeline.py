"""s_memtime stamps inside the ping-pong kernel (pf_igemm_pp.hip built with -DPP_TIMING: tools/gpu/build_variant.sh pptime -DPP_TIMING):
where do the intervals of the schedule go?  Per wavefront of the middle workgroup, for the first 12 k-steps of its second tile:
  0 MFMA phase starts | 1 MFMAs issued | 2 vmcnt(0) passed | 3 barrier passed | 4 fragment reads issued | 5 LDS-DMA pieces issued |
  6 lgkmcnt(0) passed | 7 barrier passed
Prints the differences in shader cycles.
   PF_HIP_LIB=tools/gpu/_build/libpocketflow_hip_pptime.so python tools/gpu/pp_timeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
out = torch.zeros(8 * 128, dtype=torch.int32, device='cuda')
os.environ['PF_PP_TIMING_PTR'] = hex(out.data_ptr())
os.environ['PF_IGEMM_PP'] = '2'
from pocketflow_amd import hip
B = int(os.environ.get('B', 256))
shapes = [(28, 128, 128, 3, 1), (14, 256, 256, 3, 1), (14, 1024, 256, 1, 1)]
if os.environ.get('PP_SHAPES'):
  shapes = [tuple(int(v) for v in s.split(',')) for s in os.environ['PP_SHAPES'].split(';')]
names = ['mfma issue', 'vmcnt wait', 'barrier', '-', 'reads+dma', 'waits', 'barrier']   # (group 1 waits for its LDS-DMA in 'waits', group 0 in 'vmcnt wait')
for H, C, N, k, s in shapes:
  x = torch.randn(B, H, H, C, device='cuda').bfloat16()
  w = (torch.randn(N, k, k, C, device='cuda') * 0.05).bfloat16()
  pad = (k - 1) // 2
  Ho = (H + 2 * pad - k) // s + 1
  M = B * Ho * Ho
  y = torch.empty(B, Ho, Ho, N, device='cuda', dtype=torch.bfloat16)
  G = hip.conv2d_stats_groups(M, N, geom=(B, H, H, C, N, k, k, s, pad, pad, Ho, Ho))
  partial = torch.empty(G, 4, N, device='cuda')
  for _ in range(3):
    out.zero_()
    hip.conv2d_fwd(x, w, y, B, H, H, C, N, k, k, s, pad, pad, Ho, Ho, partial=partial)
  torch.cuda.synchronize()
  t = out.cpu().numpy().astype('int64').reshape(8, 16, 8) & 0xFFFFFFFF
  print('== %d,%d,%d,%d,%d  (G = %d)' % (H, C, N, k, s, G))
  print('%-6s %-4s | %s | step' % ('wave', 'k', ' '.join('%10s' % n for n in names)))
  for wave in (0, 1, 2, 4, 5, 6):
    for ks in range(2, 8):
      row = t[wave, ks]
      if row[0] == 0:
        continue
      d = [(int(row[i + 1]) - int(row[i])) & 0xFFFFFFFF for i in range(7)]
      nxt = t[wave, ks + 1][0]
      step = ((int(nxt) - int(row[0])) & 0xFFFFFFFF) if nxt else 0
      print('%-6d %-4d | %s | %6d' % (wave, ks, ' '.join('%10d' % v for v in d), step))
