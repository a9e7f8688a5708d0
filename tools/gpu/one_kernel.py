"""Run ONE convolution kernel configuration repeatedly (for rocprofv3 --pmc / --kernel-trace passes).
  python tools/gpu/one_kernel.py igemm H C N k stride        (PF_IGEMM_TILE selects the tile)
  python tools/gpu/one_kernel.py fused HW K N res            (fused 1x1 forward: prologue + statistics [+ residual])
  python tools/gpu/one_kernel.py wrw HW K N                  (1x1 backward-filter with the prologue)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip
B = int(os.environ.get('B', 256))
IT = int(os.environ.get('IT', 12))
what = sys.argv[1]
a = [int(v) for v in sys.argv[2:]]
g = torch.Generator(device='cuda').manual_seed(1)
if what == 'igemm':
  H, C, N, k, s = a
  x = torch.randn(B, H, H, C, device='cuda', generator=g).bfloat16()
  w = (torch.randn(N, k, k, C, device='cuda', generator=g) * 0.05).bfloat16()
  pad = (k - 1) // 2
  Ho = (H + 2 * pad - k) // s + 1
  y = torch.empty(B, Ho, Ho, N, device='cuda', dtype=torch.bfloat16)
  for _ in range(IT):
    hip.conv2d_fwd(x, w, y, B, H, H, C, N, k, k, s, pad, pad, Ho, Ho)
elif what == 'fused':
  hw, K, N, res = a
  M = B * hw * hw
  X = torch.randn(M, K, device='cuda', generator=g).bfloat16()
  W = (torch.randn(N, K, device='cuda', generator=g) * 0.05).bfloat16()
  R = torch.randn(M, N, device='cuda', generator=g).bfloat16() if res else None
  ss = torch.stack([torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g)])
  slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
  hip.minmax_tensor(torch.relu(X.float() * ss[0] + ss[1]).contiguous(), slot)
  Y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  G = hip.conv1x1_stats_groups(M, N, K, prologue=True)
  partial = torch.empty(G, 4, N, device='cuda')
  for _ in range(IT):
    hip.conv1x1_fwd(X, W, Y, M, N, K, R=R, scale_shift=ss, act='Relu', slot=slot, bits=8, partial=partial)
elif what == 'wrw':
  hw, K, N = a
  M = B * hw * hw
  X = torch.randn(M, K, device='cuda', generator=g).bfloat16()
  dY = (torch.randn(M, N, device='cuda', generator=g) * 0.1).bfloat16()
  ss = torch.stack([torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g)])
  slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
  hip.minmax_tensor(torch.relu(X.float() * ss[0] + ss[1]).contiguous(), slot)
  ws = torch.empty((hip.conv1x1_wrw_splits(M, N, K) + 32) * N * K, device='cuda')
  dW = torch.empty(N, K, device='cuda', dtype=torch.bfloat16)
  for _ in range(IT):
    hip.conv1x1_wrw(dY, X, dW, ws, M, N, K, scale_shift=ss, act='Relu', slot=slot, bits=8)
torch.cuda.synchronize()
