"""Cycle stamps inside k_wrw2 (build -DPF_W2_TIMING, tools/gpu/build_ablate.sh): for steps 4..11 of the middle workgroup, per
wavefront: [0] loop top, [1] after the LDS-DMA issue of step t+2, [2] after both fragment-read batches (asm, each ends with
lgkmcnt(0)), [3] after the MFMA issue, [4] after the counted vmcnt wait, [5] after the prologue pass + lgkmcnt(0), [6] after the
barrier.  Prints the mean duration of every segment in cycles."""
import ctypes, os, sys
from ctypes import c_int, c_void_p, c_int64
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pocketflow_amd import hip

here = os.path.dirname(os.path.abspath(__file__))
prod = ctypes.CDLL(hip.lib_path())      # NOT RTLD_GLOBAL: the ablation builds' template kernels would bind to the product's
lib = ctypes.CDLL(os.path.join(here, '_build', 'libwrw_timing.so'))
LAUNCH = '_Z14pf_wrw2_launchPKvS0_PfPKfiPKjiiiiiiiiiiiiiilP12ihipStream_t'
SPLITS = '_Z14pf_wrw2_splitsiiii'
B = 256
for H, C, N, k, pro in [(14, 1024, 256, 1, 1), (14, 256, 256, 3, 0), (56, 256, 64, 1, 1)]:
  M = B * H * H
  x = torch.randn(M, C, device='cuda').bfloat16()
  dy = (torch.randn(M, N, device='cuda') * 0.1).bfloat16()
  ss = torch.stack([torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')])
  slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
  hip.minmax_tensor(torch.relu(x[:65536].float() * ss[0] + ss[1]).contiguous(), slot)
  taps = k * k
  S = getattr(prod, SPLITS)(c_int(M), c_int(N), c_int(C), c_int(taps))
  ws = torch.zeros((S + 32) * N * taps * C, device='cuda')
  p = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
  st = c_void_p(torch.cuda.current_stream().cuda_stream)
  pad = (k - 1) // 2
  args = (p(dy), p(x), p(ws), p(ss if pro else None), c_int(1), p(slot if pro else None), c_int(8), c_int(M), c_int(N), c_int(C), c_int(k), c_int(k),
          c_int(H), c_int(H), c_int(H), c_int(H), c_int(1), c_int(pad), c_int(pad), c_int(S), c_int64(M), st)
  for _ in range(3):
    assert getattr(lib, LAUNCH)(*args) == 0
  torch.cuda.synchronize()
  log = ws[:8 * 64].view(torch.int32).cpu().numpy().astype(np.int64).reshape(8, 8, 8)      # [wave][step][stamp]
  names = ['dma issue', 'frag reads', 'mfma issue', 'vmcnt wait', 'prologue+lgkm', 'barrier', '(loop back)']
  print('%d,%d,%d,%d,%d  S=%d' % (H, C, N, k, pro, S))
  for w in range(8):
    if log[w].max() == 0:
      continue
    seg = np.diff(log[w][:, :7], axis=1) & 0xFFFFFFFF                    # within a step
    step = (np.diff(log[w][:, 0]) & 0xFFFFFFFF)
    print('  wave %d: step %5.0f cyc | ' % (w, step.mean()) + '  '.join('%s %4.0f' % (n, v) for n, v in zip(names, seg.mean(0))))
  if os.environ.get('RAW'):
    print(log[0][:3])
