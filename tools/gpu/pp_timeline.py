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
# stamps of a k-step (v3 schedule, one barrier per k-step): 0 MFMA phase starts, 1 MFMAs issued, 2 own LDS-DMA wait passed (group 0),
# 3 barrier passed (group 0), 5 reads + pieces issued (group 0) / load phase issued (group 1, taken at the start of the NEXT iteration),
# 6 waits passed (group 1), 7 barrier passed (group 1)
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
  for wave in (0, 4):
    tt = [int(v) for v in t[wave, 12][:4]]
    if tt[0]:
      print('wave %d tile: pipeline fill %d | main loop %d | epilogue %d cycles' % (wave, (tt[1] - tt[0]) & 0xFFFFFFFF, (tt[2] - tt[1]) & 0xFFFFFFFF, (tt[3] - tt[2]) & 0xFFFFFFFF))
  for wave in (0, 2, 4, 6):
    base = None
    for ks in range(2, 8):
      row = t[wave, ks]
      if row[0] == 0:
        continue
      if base is None:
        base = int(min(v for v in row if v))
      ev = sorted((int(v) - base, i) for i, v in enumerate(row) if v)
      print('wave %d k %d | ' % (wave, ks) + '  '.join('%d@%d' % (i, tt) for tt, i in ev))
