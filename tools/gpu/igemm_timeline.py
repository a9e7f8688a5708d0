"""Cycle stamps inside the three-stage prologue kernel k_igemm<128,256,2,4,3,2> (build -DPF_IG_TIMING, tools/gpu/build_ablate.sh): the
THIRD tile of the middle workgroup, wavefront 0.  Stamps: tile top -> prologue stages issued -> first stage landed (counted vmcnt) ->
first transform + barrier -> k-steps 0..3 -> last k-step -> residual added -> next tile's first stage issued + C tile written ->
barrier -> row passes (statistics, stores issued) -> barrier -> (top of the next tile)."""
import ctypes, os, subprocess, sys
from ctypes import c_int, c_void_p, c_float
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pocketflow_amd import hip

here = os.path.dirname(os.path.abspath(__file__))
path = os.path.join(here, '_build', 'libig_timing.so')
sym = [l.split()[-1] for l in subprocess.run(['nm', '-D', path], capture_output=True, text=True).stdout.splitlines()
       if 'pf_igemm_conv1x1' in l and ' T ' in l][0]
lib = ctypes.CDLL(path)
fn = getattr(lib, sym)
B = int(os.environ.get('B', 256))
names = ['setup + stage issue', 'wait stage 0', 'transform + barrier', 'k-step 0', 'k-step 1', 'k-step 2', 'k-step 3',
         '(k-steps 4..)', 'residual add (wait R)', 'prefetch issue + C write', 'barrier', 'row passes + stores', 'barrier', 'to next tile top']
for H, K, N, res in [(14, 256, 1024, 1), (7, 512, 2048, 1), (14, 1024, 256, 0), (28, 512, 128, 0)]:
  M = B * H * H * (1 if N >= 1024 else 4)               # the narrow layers need >= 3 tiles per workgroup for a third tile
  g = torch.Generator(device='cuda').manual_seed(H + K + N)
  X = torch.randn(M, K, device='cuda', generator=g).bfloat16()
  W = (torch.randn(N, K, device='cuda', generator=g) * 0.05).bfloat16()
  R = torch.randn(M, N, device='cuda', generator=g).bfloat16() if res else None
  Y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  ss = torch.stack([torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g)])
  slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
  hip.minmax_tensor(torch.relu(X[:65536].float() * ss[0] + ss[1]).contiguous(), slot)
  G = hip.conv1x1_stats_groups(M, N, K, prologue=True)
  partial = torch.empty(G, 4, N, device='cuda')
  log = torch.zeros(64, dtype=torch.int32, device='cuda')
  os.environ['PF_IG_TIMING_PTR'] = str(log.data_ptr())
  p = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
  args = (p(X), p(W), p(Y), p(R), p(partial), c_void_p(0), c_void_p(0), c_void_p(0), c_float(0), c_float(0), p(ss), p(slot),
          c_float(255.0), c_float(0.0), c_float(float('inf')), c_int(M), c_int(N), c_int(K), c_int(0), c_int(0), c_int(0), c_int(0), c_int(1),
          c_void_p(torch.cuda.current_stream().cuda_stream))
  for _ in range(3):
    assert fn(*args) == 0
  torch.cuda.synchronize()
  t = log.cpu().numpy().astype(np.int64)[:15] & 0xFFFFFFFF
  print('%d,%d,%d,res=%d  M=%d  tiles/workgroup ~ %.1f' % (H, K, N, res, M, (M / 128) * (N / 256) / 256))
  if t[0] == 0:
    print('   (no third tile in the middle workgroup)')
    continue
  # stamps that were not taken (fewer than 4 k-steps ...) stay 0: fill forward so that their segment reads 0
  for i in range(1, 15):
    if t[i] == 0:
      t[i] = t[i - 1]
  d = (np.diff(t) & 0xFFFFFFFF)
  print('   tile: %d counts | ' % (t[14] - t[0]) + '  '.join('%s %d' % (n, v) for n, v in zip(names, d)))
