"""Prologue 1x1 convolutions (BN + ReLU + fake-quant on the input operand, residual + statistics in the epilogue) per ResNet-50
layer at B = 256: the three-stage kernel of round 3 (PF_IGEMM_PRO3=1) vs round 2's two-stage kernel (=0) vs the plain GEMM of
the same shape (no prologue, no epilogue options), with the HBM floor of the layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import gpu_time_us as timeit   # hipGraph replay: no host launch overhead in the numbers


B = int(os.environ.get('B', 256))
shapes = [(28, 512, 128, 0), (28, 512, 256, 0), (28, 128, 512, 1), (14, 256, 1024, 1), (14, 1024, 256, 0), (14, 1024, 512, 0), (7, 512, 2048, 1),
          (7, 2048, 512, 0), (56, 256, 64, 0), (56, 64, 256, 1)]
print('%-16s | %9s %9s %9s | floor(6.3TB/s) | TF(pro3)' % ('HW,K,N,res', 'pro3 us', 'pro2 us', 'plain us'))
for hw, K, N, res in shapes:
  M = B * hw * hw
  g = torch.Generator(device='cuda').manual_seed(hw + K + N)
  X = torch.randn(M, K, device='cuda', generator=g).bfloat16()
  W = (torch.randn(N, K, device='cuda', generator=g) * 0.05).bfloat16()
  R = torch.randn(M, N, device='cuda', generator=g).bfloat16() if res else None
  ss = torch.stack([torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g)])
  slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
  hip.minmax_tensor(torch.relu(X.float() * ss[0] + ss[1]).contiguous(), slot)
  Y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  t = {}
  for mode, pro3 in (('1', '1'), ('0', '0')):
    os.environ['PF_IGEMM_PRO3'] = pro3
    hip.tuning_reload()          # the library reads its switches once
    G = hip.conv1x1_stats_groups(M, N, K, prologue=True)
    partial = torch.empty(G, 4, N, device='cuda')
    t[mode] = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, R=R, scale_shift=ss, act='Relu', slot=slot, bits=8, partial=partial))
  os.environ.pop('PF_IGEMM_PRO3')
  hip.tuning_reload()          # the library reads its switches once
  tp = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K))
  # what the extras cost on the three-stage kernel: + statistics, + residual, + prologue without / with fake-quant
  G = hip.conv1x1_stats_groups(M, N, K, prologue=False)
  part0 = torch.empty(G, 4, N, device='cuda')
  v_stats = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, partial=part0))
  v_res = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, R=R)) if res else float('nan')
  v_pro = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, scale_shift=ss, act='Relu'))
  v_proq = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, scale_shift=ss, act='Relu', slot=slot, bits=8))
  extra = ' | plain+stats %4.0f  plain+res %4.0f  pro(no quant) %4.0f  pro(quant) %4.0f' % (v_stats, v_res, v_pro, v_proq)
  floor = (M * K + (2 if res else 1) * M * N) * 2 / 6.3e12 * 1e6
  print('%-16s | %9.0f %9.0f %9.0f | %6.0f | %5.0f' % ('%d,%d,%d,%d' % (hw, K, N, res), t['1'], t['0'], tp, floor, 2.0 * M * N * K / t['1'] * 1e-6) + extra)
