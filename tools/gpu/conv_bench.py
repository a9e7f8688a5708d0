"""Micro-benchmark of the fused 1x1-conv kernels on the ResNet-50 B=256 shapes vs the unfused chain
(pf_bn_act_quant_apply + MIOpen conv [+ torch add] [+ pf_bn_stats]) and MIOpen's backward kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from pocketflow_amd import hip

def timeit(fn, n=10):
  for _ in range(2): fn()
  torch.cuda.synchronize()
  a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n * 1e3   # us

B = int(os.environ.get('B', 256))
shapes = [(56, 64, 64), (56, 64, 256), (56, 256, 64), (28, 256, 128) , (28, 128, 512), (28, 512, 128), (14, 512, 256), (14, 256, 1024), (14, 1024, 256),
          (7, 1024, 512), (7, 512, 2048), (7, 2048, 512)]
torch.backends.cudnn.benchmark = True
print('%-18s | fwd: fused  apply+conv(+add+stats) hbm-floor | bwd-data: ours miopen | wrw: ours miopen' % 'HW,K,N')
for hw, K, N in shapes:
  M = B * hw * hw
  x4 = torch.randn(B, K, hw, hw, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
  X = x4.permute(0, 2, 3, 1).reshape(M, K)
  W = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
  W4 = W.view(N, K, 1, 1).contiguous(memory_format=torch.channels_last)
  R = torch.randn(M, N, device='cuda').bfloat16()
  ss = torch.stack([torch.rand(K, device='cuda') + 0.5, torch.randn(K, device='cuda')])
  slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
  hip.minmax_tensor(torch.relu(X.float() * ss[0] + ss[1]).contiguous(), slot)
  Y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  G = hip.conv1x1_stats_groups(M, N, K, prologue=True)
  partial = torch.empty(G, 4, N, device='cuda')
  Q = torch.empty_like(X)
  q4 = Q.view(B, hw, hw, K).permute(0, 3, 1, 2)
  t_fused = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, R=R, scale_shift=ss, act='Relu', slot=slot, bits=8, partial=partial))
  t_plain = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K))
  if os.environ.get('VARIANTS'):
    t_pro = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, scale_shift=ss, act='Relu', slot=slot, bits=8))
    t_pronq = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, scale_shift=ss, act='Relu'))
    t_res = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, R=R))
    t_st = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, partial=partial))
    print('   variants: plain %.0f | +prologue(quant) %.0f | +prologue(no quant) %.0f | +residual %.0f | +stats %.0f | all %.0f' % (t_plain, t_pro, t_pronq, t_res, t_st, t_fused))
  nblk = 64
  part2 = torch.empty(nblk * 4 * N, device='cuda')
  def unfused():
    hip.bn_act_quant_apply(X, Q, M, K, ss, 'Relu', slot, 8, True)
    y = F.conv2d(q4, W4)
    y = y + R.view(B, hw, hw, N).permute(0, 3, 1, 2)
    hip.bn_stats(y, M, N, part2, nblk)
  NOMI = bool(os.environ.get('NO_MIOPEN'))
  t_unf = 0 if NOMI else timeit(unfused)
  t_mi = 0 if NOMI else timeit(lambda: F.conv2d(q4, W4))
  floor = (M * K + 2 * M * N) * 2 / 8e12 * 1e6
  # backward data
  dY = torch.randn(M, N, device='cuda').bfloat16()
  dy4 = dY.view(B, hw, hw, N).permute(0, 3, 1, 2)
  Wt = W.t().contiguous()
  dX = torch.empty(M, K, device='cuda', dtype=torch.bfloat16)
  t_bd = timeit(lambda: hip.conv1x1_fwd(dY, Wt, dX, M, K, N))
  t_bd_mi = 0 if NOMI else timeit(lambda: torch.ops.aten.convolution_backward(dy4, q4, W4, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
  ws = torch.empty((hip.conv1x1_wrw_splits(M, N, K) + 32) * N * K, device='cuda')
  dW = torch.empty(N, K, device='cuda', dtype=torch.bfloat16)
  t_wr = timeit(lambda: hip.conv1x1_wrw(dY, X, dW, ws, M, N, K, scale_shift=ss, act='Relu', slot=slot, bits=8))
  t_wr_mi = 0 if NOMI else timeit(lambda: torch.ops.aten.convolution_backward(dy4, q4, W4, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
  print('%-18s | %7.0f (plain %5.0f) %7.0f (conv %5.0f) %6.0f | %7.0f %7.0f | %7.0f %7.0f' % (
      '%d,%d,%d' % (hw, K, N), t_fused, t_plain, t_unf, t_mi, floor, t_bd, t_bd_mi, t_wr, t_wr_mi))
