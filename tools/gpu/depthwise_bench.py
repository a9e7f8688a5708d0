"""Per-layer timings of the depthwise kernels (pf_depthwise.hip) on MobileNet-v1's thirteen depthwise layers at B = 256, bf16:
forward (+ BN statistics), backward-data, backward-filter with the one-thread-per-output slab reduction (default) and with the
staged reduction, each against its HBM floor (bytes / 4.5 TB/s) and against torch's grouped convolution (MIOpen).
Also MobileNet's image convolution (3 -> 32, 3x3 stride 2) on k_convg against MIOpen and its floor.
Written at the end of round 4 from the step table (profiles/r04_step_kernels_c3.csv: k_dw_wrw_reduce 271 us per launch,
k_dw_fwd 163 us average); not yet run.

    python tools/gpu/depthwise_bench.py            (B=..., DEPTH_MULT=1.0 from the environment)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from pocketflow_amd import hip
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import gpu_time_us as bench_us      # noqa: E402  (tools/gpu/_timing.py)

B = int(os.environ.get('B', 256))
DM = float(os.environ.get('DEPTH_MULT', '1.0'))
BW = 4.5e12                                      # what a streaming kernel reaches on this part (k_bn_bwd_apply: 4.6 TB/s)
# (input size, channels at depth multiplier 1, stride) of the depthwise layers, utils/external/mobilenet_v1.py _CONV_DEFS
LAYERS = [(112, 32, 1), (112, 64, 2), (56, 128, 1), (56, 128, 2), (28, 256, 1), (28, 256, 2), (14, 512, 1), (14, 512, 1),
          (14, 512, 1), (14, 512, 1), (14, 512, 1), (14, 512, 2), (7, 1024, 1)]


def same(size, stride):
  out = -(-size // stride)
  total = max((out - 1) * stride + 3 - size, 0)
  return total // 2, out


print('%-20s | %8s %8s | %8s %8s | %8s %8s %8s | %8s %8s %8s' % ('H, C, stride', 'fwd', 'floor', 'bwd-data', 'floor', 'wrw', '-', 'floor',
                                                                'mi fwd', 'mi bwd', 'mi wrw'))
tot = [0.0] * 10
for H, C1, stride in LAYERS:
  C = max(8, int(C1 * DM))
  if not hip.depthwise_supported(C, 3, stride):
    print('%-20s | not supported' % ('%d, %d, %d' % (H, C, stride)))
    continue
  ph, Ho = same(H, stride)
  g = torch.Generator(device='cuda').manual_seed(H + C + stride)
  x = torch.randn(B, C, H, H, device='cuda', generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
  w = (torch.randn(C, 3, 3, device='cuda', generator=g) * 0.3).bfloat16()
  y = torch.empty(B, C, Ho, Ho, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
  dy = torch.randn(B, C, Ho, Ho, device='cuda', generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
  dx = torch.empty_like(x)
  dw = torch.empty(C, 3, 3, device='cuda', dtype=torch.bfloat16)
  G = hip.depthwise_groups(B, Ho, Ho, C)
  partial = torch.empty(G, 4, C, device='cuda')
  slabs = torch.empty((G + 32) * C * 9, device='cuda')
  t_f = bench_us(lambda: hip.depthwise_fwd(x, w, y, B, H, H, C, 3, stride, ph, ph, Ho, Ho, partial=partial))
  t_b = bench_us(lambda: hip.depthwise_bwd_data(dy, w, dx, B, H, H, C, 3, stride, ph, ph, Ho, Ho))
  t_w = bench_us(lambda: hip.depthwise_wrw(dy, x, dw, slabs, B, H, H, C, 3, stride, ph, ph, Ho, Ho))
  t_w2, err = t_w, 0.0                                    # (round 4's per-output slab reduction is gone: one form since round 5)
  n_in, n_out = B * H * H * C * 2, B * Ho * Ho * C * 2
  fl = (n_in + n_out) / BW * 1e6
  fl_w = (n_in + n_out + G * C * 9 * 4 * 2) / BW * 1e6
  # torch reference (asymmetric 'SAME' pads of the strided layers as an explicit pad: MIOpen sees a VALID convolution)
  pe = max((Ho - 1) * stride + 3 - H - ph, 0)
  xp = F.pad(x, (ph, pe, ph, pe)).contiguous(memory_format=torch.channels_last)
  w4 = w.reshape(C, 1, 3, 3)
  m_f = bench_us(lambda: F.conv2d(xp, w4, stride=stride, groups=C))
  m_b = bench_us(lambda: torch.ops.aten.convolution_backward(dy, xp, w4, None, [stride, stride], [0, 0], [1, 1], False, [0, 0], C, [True, False, False]))
  m_w = bench_us(lambda: torch.ops.aten.convolution_backward(dy, xp, w4, None, [stride, stride], [0, 0], [1, 1], False, [0, 0], C, [False, True, False]))
  print('%-20s | %8.1f %8.1f | %8.1f %8.1f | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f   (G = %d, red2 vs default: %.1e)' % (
      '%d, %d, %d' % (H, C, stride), t_f, fl, t_b, fl, t_w, t_w2, fl_w, m_f, m_b, m_w, G, err))
  for i, v in enumerate((t_f, fl, t_b, fl, t_w, t_w2, fl_w, m_f, m_b, m_w)):
    tot[i] += v
print('%-20s | %8.1f %8.1f | %8.1f %8.1f | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f' % (('sum',) + tuple(tot)))

# MobileNet's first convolution: 3 -> 32 * DM channels, 3x3 stride 2, 'SAME' (pad 0 in front, 1 behind at 224)
N = max(8, int(32 * DM))
H, Ho = 224, 112
x = torch.randn(B, 3, H, H, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(N, 3, 3, 3, device='cuda') * 0.1).bfloat16()                  # KRSC memory
y = torch.empty(B, N, Ho, Ho, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
dy = torch.randn(B, N, Ho, Ho, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
dwk = torch.empty(N, 3, 3, 3, device='cuda', dtype=torch.bfloat16)
slab = torch.empty(hip.convg_wrw_splits(B, 3, N, 3, 3, Ho, Ho) * N * 27, device='cuda')
t_f = bench_us(lambda: hip.convg_fwd(x, w, None, y, B, H, H, 3, N, 3, 3, 2, 0, 0, Ho, Ho))
t_w = bench_us(lambda: hip.convg_wrw(dy, x, dwk, slab, B, H, H, 3, N, 3, 3, 2, 0, 0, Ho, Ho))
xp = F.pad(x, (0, 1, 0, 1)).contiguous(memory_format=torch.channels_last)
w4 = w.permute(0, 3, 1, 2)
m_f = bench_us(lambda: F.conv2d(xp, w4, stride=2))
m_w = bench_us(lambda: torch.ops.aten.convolution_backward(dy, xp, w4, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
# the same with the channels padded to 4 (k_convg's 4-element vector loader; measured as PF_CONVG_PAD_C3 in round 5: +0.1 % per C3 step, not kept), pad copies included
def padded_fwd():
  x4 = torch.zeros(B, 4, H, H, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
  x4[:, :3] = x
  w4p = torch.zeros(N, 3, 3, 4, device='cuda', dtype=torch.bfloat16)
  w4p[..., :3] = w
  hip.convg_fwd(x4, w4p, None, y, B, H, H, 4, N, 3, 3, 2, 0, 0, Ho, Ho)
  return x4


x4 = padded_fwd()
dwk4 = torch.empty(N, 3, 3, 4, device='cuda', dtype=torch.bfloat16)
slab4 = torch.empty(hip.convg_wrw_splits(B, 4, N, 3, 3, Ho, Ho) * N * 36, device='cuda')
p_f = bench_us(padded_fwd)
p_w = bench_us(lambda: hip.convg_wrw(dy, x4, dwk4, slab4, B, H, H, 4, N, 3, 3, 2, 0, 0, Ho, Ho))
print('  padded to 4 channels: forward (with the pad copies) %.1f us, backward-filter %.1f us; dW agrees to %.1e' % (
    p_f, p_w, float((dwk4[..., :3].float() - dwk.float()).abs().max() / (dwk.float().abs().max() + 1e-12))))
fl = (B * H * H * 3 * 2 + B * Ho * Ho * N * 2) / BW * 1e6
print('image convolution 3 -> %d, 3x3 / 2 at %d: k_convg forward %.1f us, backward-filter %.1f us; MIOpen %.1f / %.1f; floor %.1f' % (
    N, H, t_f, t_w, m_f, m_w, fl))
