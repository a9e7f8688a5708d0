"""The index arithmetic of pf_conv2d_bwd_data_strided (csrc/pf_igemm.hip), restated in Python and checked against autograd on the CPU:
backward-data of a stride-st convolution = st * st stride-1 convolutions over dY, one per output-parity class, each walking a
sub-grid of the flipped / transposed kernel buffer Wt[c][R-1-r][S-1-s][n] and scattering its rows to the class's pixels.  The GPU
test of the kernel itself is tests/test_igemm_gpu.py::test_conv2d_backward_data_of_strided_convolutions_by_parity_classes; this file
pins the host-side formulas (first tap, tap count, begin pad, first flipped row and its step) that the launcher computes per class."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def class_plan(a, pad, R, stride):
  """Per axis, for output-parity class a: (taps th, begin pad of the stride-1 convolution over dY, first flipped kernel row, row step)
  -- the same expressions as the launcher's (r1, th, dmin, w_r0, w_rs)."""
  r1 = (a + pad) % stride
  th = (R - r1 + stride - 1) // stride
  dmin = (a + pad - r1) // stride - (th - 1)
  w_r0 = (R - 1 - r1) - (th - 1) * stride
  return th, -dmin, w_r0, stride


@pytest.mark.parametrize('H,W,C,N,R,S,stride,pad', [(8, 8, 3, 4, 3, 3, 2, 1), (12, 10, 2, 3, 5, 5, 2, 2), (8, 8, 2, 2, 2, 2, 2, 0),
                                                     (9, 9, 2, 3, 3, 3, 3, 1), (12, 8, 2, 2, 7, 7, 2, 3), (8, 12, 3, 2, 3, 5, 2, 1)])
def test_parity_class_decomposition_equals_autograd(H, W, C, N, R, S, stride, pad):
  g = torch.Generator().manual_seed(H * 31 + R)
  Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
  x = torch.randn(2, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
  w = torch.randn(N, C, R, S, generator=g, dtype=torch.float64)
  dy = torch.randn(2, N, Ho, Wo, generator=g, dtype=torch.float64)
  F.conv2d(x, w, stride=stride, padding=pad).backward(dy)
  # Wt[c][r'][s'][n] = W[n][c][R-1-r'][S-1-s']  (VarStore.transposed: flipped, transposed)
  wt = w.flip(2, 3).permute(1, 2, 3, 0)
  dx = torch.full((2, C, H, W), float('nan'), dtype=torch.float64)
  Hc, Wc = H // stride, W // stride
  assert H % stride == 0 and W % stride == 0 and R >= stride and S >= stride      # the launcher's preconditions
  for ay in range(stride):
    th, ph, r0, rs = class_plan(ay, pad, R, stride)
    for ax in range(stride):
      tw, pw, s0, ss = class_plan(ax, pad, S, stride)
      sub = wt[:, r0:r0 + th * rs:rs, s0:s0 + tw * ss:ss, :]                        # [C][th][tw][N]: the sub-grid walked in place
      assert sub.shape[1] == th and sub.shape[2] == tw
      # stride-1 convolution over dY with begin pads (ph, pw); positions beyond dY read zeros; output grid Hc x Wc
      need_h, need_w = Hc - 1 + th - ph, Wc - 1 + tw - pw                           # last dY row / column + 1 that is touched
      dyp = F.pad(dy, (max(pw, 0), max(need_w - Wo, 0), max(ph, 0), max(need_h - Ho, 0)))
      off_h, off_w = max(-ph, 0), max(-pw, 0)                                        # negative begin pad = the window starts inside dY
      out = F.conv2d(dyp[:, :, off_h:, off_w:], sub.permute(0, 3, 1, 2))[:, :, :Hc, :Wc]
      assert out.shape[2] == Hc and out.shape[3] == Wc
      dx[:, :, ay::stride, ax::stride] = out                                         # the scatter: row (i, j) -> pixel (st*i + ay, st*j + ax)
  assert torch.isfinite(dx).all()                                                    # every pixel written exactly once
  np.testing.assert_allclose(dx.numpy(), x.grad.numpy(), rtol=1e-10, atol=1e-10)
