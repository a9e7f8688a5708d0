"""Round-6 experiment: 8-bit activation CODES as the input operand of the forward implicit GEMM, against the bf16 operand of the
product kernel, on the 3x3 layers the step dispatches to k_igemm.  With uql_activation_bits = 8 every materialised post-BN-ReLU
activation lies on a 256-level grid x = alpha * c; storing c (one byte) halves the operand's bytes through HBM / L2 / LDS-DMA /
the fragment reads and is exact, where bf16 storage of alpha * c is not.  The codes kernel is k_igemm itself compiled with
-DPF_IG_CODES into a variant library (tools/gpu/build_variant.sh codes -DPF_IG_CODES): same tiles, same ring, same epilogue.

  bash tools/gpu/build_variant.sh codes -DPF_IG_CODES && python tools/gpu/a8_codes_probe.py  ->  profiles/r06_a8_codes_ab.txt
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F
from pocketflow_amd import hip
from _timing import gpu_time_us

here = os.path.dirname(os.path.abspath(__file__))
var = ctypes.CDLL(os.path.join(here, '_build', 'libpocketflow_hip_codes.so'))
B = int(os.environ.get('B', 256))
shapes = [(14, 256, 256, 3, 1), (28, 128, 128, 3, 1), (7, 512, 512, 3, 1), (28, 256, 256, 3, 2), (14, 1024, 256, 1, 1)]
print('batch %d; alpha = 6/255; codes ~ relu(N(20, 40)) rounded, clipped to 0..255; kernels 0.05 * N(0, 1) in bf16; errors against a float64' % B)
print('convolution of the EXACT grid values over the first 4 images (both kernels round their output to bf16: |y| * 2^-9 at most)')
print('%-18s | %-28s | %-28s | time ratio' % ('H,C,N,k,s', 'bf16 operand (product)', 'code operand (variant)'))
print('%-18s | %8s %9s %9s | %8s %9s %9s |' % ('', 'us', 'max err', 'mean err', 'us', 'max err', 'mean err'))
for H, C, N, k, s in shapes:
  g = torch.Generator(device='cuda').manual_seed(H + C + N)
  codes = (torch.randn(B, H, H, C, device='cuda', generator=g) * 40 + 20).round().clamp(0, 255).to(torch.uint8)
  alpha = 6.0 / 255.0
  xb = (codes.float() * alpha).bfloat16()                       # what k_bn_apply stores today
  w = (torch.randn(N, k, k, C, device='cuda', generator=g) * 0.05).bfloat16()
  pad = (k - 1) // 2
  Ho = (H + 2 * pad - k) // s + 1
  y0 = torch.empty(B, Ho, Ho, N, device='cuda', dtype=torch.bfloat16)
  y1 = torch.empty_like(y0)
  zero = hip.zero_page(xb.device)
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

  def f_prod():
    hip.conv2d_fwd(xb, w, y0, B, H, H, C, N, k, k, s, pad, pad, Ho, Ho)

  def f_codes():
    stc = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    r = var.pf_probe_conv2d_fwd_codes(ctypes.c_void_p(codes.data_ptr()), ctypes.c_float(alpha), ctypes.c_void_p(w.data_ptr()),
                                      ctypes.c_void_p(y1.data_ptr()), ctypes.c_void_p(zero.data_ptr()), B, H, H, C, N, k, k, s, pad,
                                      pad, Ho, Ho, stc)
    assert r == 0, r
  f_prod(); f_codes()
  torch.cuda.synchronize()
  n = 4
  ref = F.conv2d((codes[:n].double() * alpha).permute(0, 3, 1, 2).cpu(), w[:].double().permute(0, 3, 1, 2).cpu(), stride=s,
                 padding=pad).permute(0, 2, 3, 1)
  e0 = (y0[:n].double().cpu() - ref).abs()
  e1 = (y1[:n].double().cpu() - ref).abs()
  whole = float((y0.float() - y1.float()).abs().max())          # the whole batch: the two kernels against each other
  t0 = gpu_time_us(f_prod)
  t1 = gpu_time_us(f_codes)
  print('%-18s | %8.1f %9.2e %9.2e | %8.1f %9.2e %9.2e | %.3f   (max |y_bf16 - y_codes| over the batch %.2e, max |y| %.1f)' % (
      '%d,%d,%d,%d,%d' % (H, C, N, k, s), t0, float(e0.max()), float(e0.mean()), t1, float(e1.max()), float(e1.mean()), t1 / t0, whole,
      float(ref.abs().max())))
