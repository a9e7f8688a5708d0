"""Per-layer timings of round 4's kernels at the bench shapes (B = 256, bf16): strided backward-data by parity classes vs MIOpen, the
dense layer on k_convg (with / without the split contraction) vs rocBLAS, the 3x3 backward-filter at C = 64 vs MIOpen."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from pocketflow_amd import hip
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import gpu_time_us as bench_us      # noqa: E402  (tools/gpu/_timing.py)

B = int(os.environ.get('B', 256))
print('strided 3x3 backward-data (stride 2, pad 1), us                    ours (4 launches)   MIOpen')
for H, C in ((56, 128), (28, 256), (14, 512)):
  N = C
  Ho = H // 2
  dy = torch.randn(B, Ho, Ho, N, device='cuda').bfloat16()
  w = (torch.randn(N, 3, 3, C, device='cuda') * 0.05).bfloat16()
  wb = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
  dx = torch.empty(B, H, H, C, device='cuda', dtype=torch.bfloat16)
  ours = bench_us(lambda: hip.conv2d_bwd_data_strided(dy, wb, dx, B, H, H, C, N, 3, 3, 2, 1, 1, Ho, Ho))
  x_ = torch.empty(B, C, H, H, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
  dy_ = dy.permute(0, 3, 1, 2)
  w_ = w.permute(0, 3, 1, 2)
  mi = bench_us(lambda: torch.ops.aten.convolution_backward(dy_, x_, w_, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
  gf = 2.0 * B * Ho * Ho * N * C * 9 / 1e9
  print('  %3dx%-3d %4d -> %-4d  %8.1f GFLOP            %8.1f (%4.0f TF/s)   %8.1f' % (H, H, C, N, gf, ours, gf / ours * 1e3, mi))
print('dense 2048 -> 1001 at B = %d, us                                   k_convg   k_convg split   rocBLAS' % B)
x = torch.randn(B, 2048, device='cuda').bfloat16()
w = (torch.randn(1001, 2048, device='cuda') * 0.02).bfloat16()
bias = torch.zeros(1001, device='cuda')
y = torch.empty(B, 1001, device='cuda', dtype=torch.bfloat16)
ws = torch.empty(1 << 22, device='cuda')
a = bench_us(lambda: hip.convg_fwd(x, w, bias, y, B, 1, 1, 2048, 1001, 1, 1, 1, 0, 0, 1, 1))
b = bench_us(lambda: hip.convg_fwd(x, w, bias, y, B, 1, 1, 2048, 1001, 1, 1, 1, 0, 0, 1, 1, slab=ws))
c = bench_us(lambda: F.linear(x, w, bias.bfloat16()))
print('  forward        %8.1f %8.1f %8.1f' % (a, b, c))
dy = torch.randn(B, 1001, device='cuda').bfloat16()
dx = torch.empty(B, 2048, device='cuda', dtype=torch.bfloat16)
a = bench_us(lambda: hip.convg_bwd_data(dy, w, dx, B, 1, 1, 2048, 1001, 1, 1, 1, 0, 0, 1, 1))
b = bench_us(lambda: hip.convg_bwd_data(dy, w, dx, B, 1, 1, 2048, 1001, 1, 1, 1, 0, 0, 1, 1, slab=ws))
c = bench_us(lambda: dy @ w)
print('  backward-data  %8.1f %8.1f %8.1f' % (a, b, c))
dw = torch.empty(1001, 1, 1, 2048, device='cuda', dtype=torch.bfloat16)
slab = torch.empty(hip.convg_wrw_splits(B, 2048, 1001, 1, 1, 1, 1) * 1001 * 2048, device='cuda')
a = bench_us(lambda: hip.convg_wrw(dy, x, dw, slab, B, 1, 1, 2048, 1001, 1, 1, 1, 0, 0, 1, 1))
c = bench_us(lambda: dy.t() @ x)
print('  backward-filter %7.1f %8s %8.1f' % (a, '-', c))
