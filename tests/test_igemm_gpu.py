"""Implicit-GEMM convolution kernel (pf_conv2d_fwd, pf_igemm.hip) against float32 torch references built from the SAME
bf16 inputs: 3x3 / 1x1 / 5x3 windows, strides 1 and 2, zero padding at the borders, row / channel tails, every tile
configuration, the statistics epilogue, the BN-backward-statistics epilogue, and backward-data through the flipped kernel."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
  from pocketflow_amd import hip as h
  return h


def _bf(x):
  return x.to(torch.bfloat16)


def _close(got, ref, what, frac_tol=0.0):
  got, ref = got.float(), ref.float()
  err = (got - ref).abs()
  tol = ref.abs() * 2 ** -7 + 2e-2 * float(ref.abs().mean() + 1e-6)
  bad = float((err > tol).float().mean())
  assert bad <= frac_tol, '%s: %.3e of the elements differ (max err %.3e, mean |ref| %.3e)' % (
      what, bad, float(err.max()), float(ref.abs().mean()))


def _run(hip, x_nhwc, w_krsc, stride, pad, **kw):
  imgs, H, Wd, C = x_nhwc.shape
  N, th, tw, _ = w_krsc.shape
  Ho = (H + 2 * pad[0] - th) // stride + 1
  Wo = (Wd + 2 * pad[1] - tw) // stride + 1
  y = torch.empty(imgs, Ho, Wo, N, device='cuda', dtype=torch.bfloat16)
  hip.conv2d_fwd(x_nhwc, w_krsc, y, imgs, H, Wd, C, N, th, tw, stride, pad[0], pad[1], Ho, Wo, **kw)
  return y


def _ref(x_nhwc, w_krsc, stride, pad):
  return F.conv2d(x_nhwc.float().permute(0, 3, 1, 2), w_krsc.float().permute(0, 3, 1, 2), stride=stride,
                  padding=pad).permute(0, 2, 3, 1)


# tile '': the dispatcher's own choice; every other value is a tile override (PF_IGEMM_TILE) of the per-tap implicit GEMM
@pytest.mark.parametrize('tile', ['', '256x128', '128x128', '256x64', '128x64'])
@pytest.mark.parametrize('imgs,H,Wd,C,N,k,stride', [(2, 14, 14, 64, 64, 3, 1), (3, 9, 11, 128, 128, 3, 1),
                                                    (2, 16, 16, 64, 128, 3, 2), (5, 7, 7, 192, 256, 1, 1),
                                                    (1, 20, 12, 64, 72, 3, 1), (40, 28, 28, 128, 128, 3, 1),
                                                    (37, 7, 7, 512, 256, 3, 1), (3, 56, 56, 64, 64, 3, 1), (5, 5, 62, 64, 192, 3, 1)])
def test_conv2d_fwd_matches_torch(hip, monkeypatch, tile, imgs, H, Wd, C, N, k, stride):
  if tile:
    monkeypatch.setenv('PF_IGEMM_TILE', tile)
  else:
    monkeypatch.delenv('PF_IGEMM_TILE', raising=False)
  g = torch.Generator(device='cuda').manual_seed(H * Wd + C + N)
  x = _bf(torch.randn(imgs, H, Wd, C, device='cuda', generator=g))
  w = _bf(torch.randn(N, k, k, C, device='cuda', generator=g) * 0.05)
  pad = ((k - 1) // 2, (k - 1) // 2)
  y = _run(hip, x, w, stride, pad)
  _close(y, _bf(_ref(x, w, stride, pad)), 'fwd %s' % tile)


def test_conv2d_fwd_asymmetric_window_and_padding(hip):
  """th != tw, pad_h != pad_w, an ASYMMETRIC kernel (a transposed window or tap order would show)."""
  g = torch.Generator(device='cuda').manual_seed(1)
  x = _bf(torch.randn(2, 13, 10, 64, device='cuda', generator=g))
  w = _bf(torch.randn(64, 5, 3, 64, device='cuda', generator=g) * 0.05)
  y = _run(hip, x, w, 1, (2, 0))
  _close(y, _bf(_ref(x, w, 1, (2, 0))), 'asymmetric')


@pytest.mark.parametrize('imgs,H,C,N', [(8, 28, 128, 128), (4, 14, 256, 256), (16, 56, 64, 64), (70, 14, 128, 256), (300, 7, 64, 128)])
def test_conv2d_fwd_statistics_and_residual(hip, imgs, H, C, N):
  g = torch.Generator(device='cuda').manual_seed(C)
  x = _bf(torch.randn(imgs, H, H, C, device='cuda', generator=g))
  w = _bf(torch.randn(N, 3, 3, C, device='cuda', generator=g) * 0.05)
  M = imgs * H * H
  G = hip.conv2d_stats_groups(M, N, geom=(imgs, H, H, C, N, 3, 3, 1, 1, 1, H, H))
  partial = torch.full((G, 4, N), float('nan'), device='cuda')
  r = _bf(torch.randn(M, N, device='cuda', generator=g))
  y = _run(hip, x, w, 1, (1, 1), R=r, partial=partial)
  ref = _bf(_ref(x, w, 1, (1, 1)).float().reshape(M, N) + r.float())       # fp32 sum, ONE rounding (accumulator-level add)
  _close(y.reshape(M, N), ref, 'residual')
  yf = y.float().reshape(M, N)
  assert not torch.isnan(partial).any()
  torch.testing.assert_close(partial[:, 0].sum(0), yf.sum(0), rtol=1e-4, atol=2e-2)
  torch.testing.assert_close(partial[:, 1].sum(0), (yf * yf).sum(0), rtol=1e-4, atol=2e-2)
  assert torch.equal(partial[:, 2].min(0).values, yf.min(0).values)
  assert torch.equal(partial[:, 3].max(0).values, yf.max(0).values)
  # deterministic (fixed-order reductions)
  p2 = torch.empty_like(partial)
  _run(hip, x, w, 1, (1, 1), R=r, partial=p2)
  assert torch.equal(partial, p2)


def test_conv2d_backward_data_through_flipped_kernel_with_bn_statistics(hip):
  """dX = conv(dY, W') with W'[c][r][s][n] = W[n][2-r][2-s][c], pad 1 (stride-1 3x3), against autograd; the BN-backward
  sums of the producer BN in the epilogue against pf_bn_bwd_stats on the stored dX."""
  imgs, H, C, N = 6, 14, 128, 192
  g = torch.Generator(device='cuda').manual_seed(9)
  x = _bf(torch.randn(imgs, H, H, C, device='cuda', generator=g))
  w = _bf(torch.randn(N, 3, 3, C, device='cuda', generator=g) * 0.05)
  dy = _bf(torch.randn(imgs, H, H, N, device='cuda', generator=g) * 0.1)
  xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
  F.conv2d(xt, w.float().permute(0, 3, 1, 2), padding=1).backward(dy.float().permute(0, 3, 1, 2))
  ref = xt.grad.permute(0, 2, 3, 1)
  wb = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()            # [C][3][3][N]
  M = imgs * H * H
  bnx = _bf(torch.randn(M, C, device='cuda', generator=g))
  ss = torch.stack([torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g) * 0.3])
  mi = torch.stack([torch.randn(C, device='cuda', generator=g) * 0.1, torch.rand(C, device='cuda', generator=g) + 0.5])
  G = hip.conv2d_stats_groups(M, C, geom=(imgs, H, H, N, C, 3, 3, 1, 1, 1, H, H))
  partial = torch.full((G, 2, C), float('nan'), device='cuda')
  dx = _run(hip, dy, wb, 1, (1, 1), partial=partial, bn_x=bnx, bn_scale_shift=ss, bn_mean_invstd=mi, bn_act='Relu')
  _close(dx, _bf(ref), 'bwd-data')
  nblk = 16
  ref_partial = torch.empty(nblk * 2 * C, device='cuda')
  hip.bn_bwd_stats(dx.reshape(M, C), bnx, M, C, ss, mi, 'Relu', ref_partial, nblk)
  dgamma, dbeta = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
  hip.bn_bwd_finalize(ref_partial, nblk, C, dgamma, dbeta)
  dg2, db2 = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
  hip.bn_bwd_finalize(partial, G, C, dg2, db2)
  torch.testing.assert_close(db2, dbeta, rtol=1e-4, atol=1e-3)
  torch.testing.assert_close(dg2, dgamma, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('imgs,H,C,N,k,stride,dw_dtype', [(8, 28, 128, 128, 3, 1, torch.float32), (12, 14, 256, 256, 3, 1, torch.bfloat16),
                                                          (16, 56, 64, 64, 3, 1, torch.float32), (16, 28, 128, 128, 3, 2, torch.float32),
                                                          (10, 30, 64, 192, 1, 2, torch.float32)])
@pytest.mark.parametrize('impl', ['shared-tile', 'wave-private'])
def test_conv2d_wrw_matches_autograd(hip, monkeypatch, imgs, H, C, N, k, stride, dw_dtype, impl):
  monkeypatch.setenv('PF_WRW2', '1' if impl == 'shared-tile' else '0')
  """Backward-filter on the transposed-LDS-read kernel (pf_conv2d_wrw): 3x3 stride 1 / 2 with zero padding at the borders
  (padding taps must contribute exactly 0), KRSC output, deterministic."""
  g = torch.Generator(device='cuda').manual_seed(C + N + H)
  pad = (k - 1) // 2
  x = _bf(torch.randn(imgs, H, H, C, device='cuda', generator=g))
  Ho = (H + 2 * pad - k) // stride + 1
  dy = _bf(torch.randn(imgs, Ho, Ho, N, device='cuda', generator=g) * 0.1)
  M = imgs * Ho * Ho
  S = hip.conv2d_wrw_splits(M, N, C, k * k)
  assert S > 0
  ws = torch.empty((S + 32) * N * k * k * C, device='cuda')
  dw = torch.empty(N, k, k, C, device='cuda', dtype=dw_dtype)
  hip.conv2d_wrw(dy, x, dw, ws, imgs, H, H, C, N, k, k, stride, pad, pad, Ho, Ho)
  w0 = torch.zeros(N, C, k, k, device='cuda')
  ref = torch.ops.aten.convolution_backward(dy.float().permute(0, 3, 1, 2), x.float().permute(0, 3, 1, 2), w0, None,
                                            [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
  ref = ref.permute(0, 2, 3, 1)
  torch.testing.assert_close(dw.float(), ref, rtol=2e-2 if dw_dtype == torch.bfloat16 else 2e-3, atol=3e-2)
  dw2 = torch.empty_like(dw)
  hip.conv2d_wrw(dy, x, dw2, ws, imgs, H, H, C, N, k, k, stride, pad, pad, Ho, Ho)
  assert torch.equal(dw, dw2)


# ---- the ResNet stem (pf_stem.hip) ------------------------------------------------------------------------------------
@pytest.mark.parametrize('imgs,H,Wd', [(3, 64, 64), (2, 224, 224), (1, 38, 96), (5, 18, 32), (130, 32, 64)])
def test_conv_stem_fwd_matches_torch(hip, imgs, H, Wd):
  """7x7 / stride 2 / pad 3, 3 -> 64 channels: borders on all four sides (zero padding), a partial last strip of output
  rows (H/2 not a multiple of 8), more (image, strip) items than persistent workgroups, an ASYMMETRIC random kernel (a
  transposed window, a wrong tap or channel order would show), several calls into the same output buffer."""
  assert hip.conv_stem_supported(H, Wd, 3, 64, 7, 2, 3) and not hip.conv_stem_supported(H, Wd, 4, 64, 7, 2, 3)
  g = torch.Generator(device='cuda').manual_seed(H + Wd + imgs)
  x = _bf(torch.randn(imgs, H, Wd, 3, device='cuda', generator=g))
  w = _bf(torch.randn(64, 7, 7, 3, device='cuda', generator=g) * 0.1)
  ref = _bf(_ref(x, w, 2, (3, 3)))
  for rep in range(2):
    y = torch.full((imgs, H // 2, Wd // 2, 64), float('nan'), device='cuda', dtype=torch.bfloat16)
    hip.conv_stem_fwd(x, w, y, imgs, H, Wd)
    _close(y, ref, 'stem %dx%dx%d call %d' % (imgs, H, Wd, rep))
  # one-hot probes: every (tap, channel) weight must meet exactly its input element
  x1 = torch.zeros(1, 32, 32, 3, device='cuda', dtype=torch.bfloat16)
  x1[0, 10, 13, 1] = 1.0
  y1 = torch.empty(1, 16, 16, 64, device='cuda', dtype=torch.bfloat16)
  hip.conv_stem_fwd(x1, w, y1, 1, 32, 32)
  assert torch.equal(y1, _bf(_ref(x1, w, 2, (3, 3))))


@pytest.mark.parametrize('dw_dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('imgs,H,Wd', [(3, 64, 64), (2, 224, 224), (1, 38, 96), (5, 18, 32), (130, 32, 64)])
def test_conv_stem_wrw_matches_autograd(hip, imgs, H, Wd, dw_dtype):
  """Backward-filter of the stem against aten's float32 convolution_backward on the same bf16 operands: an odd number of
  output rows (a half-empty last strip), more (image, strip) items than persistent workgroups, border taps; float32 and
  bf16 gradient buffers; bit-reproducible (fixed-order slab reduction)."""
  S = hip.conv_stem_wrw_slabs(imgs, H, Wd)
  assert S > 0 and hip.conv_stem_wrw_slabs(imgs, H, 512) == 0
  g = torch.Generator(device='cuda').manual_seed(H * 3 + Wd + imgs)
  x = _bf(torch.randn(imgs, H, Wd, 3, device='cuda', generator=g))
  dy = _bf(torch.randn(imgs, H // 2, Wd // 2, 64, device='cuda', generator=g) * 0.1)
  ref = torch.ops.aten.convolution_backward(dy.float().permute(0, 3, 1, 2), x.float().permute(0, 3, 1, 2),
                                            torch.zeros(64, 3, 7, 7, device='cuda'), None, [2, 2], [3, 3], [1, 1], False,
                                            [0, 0], 1, [False, True, False])[1].permute(0, 2, 3, 1)
  outs = []
  for rep in range(2):
    ws = torch.full(((S + 32) * 64 * 147,), float('nan'), device='cuda')
    dw = torch.full((64, 7, 7, 3), float('nan'), device='cuda', dtype=dw_dtype)
    hip.conv_stem_wrw(dy, x, dw, ws, imgs, H, Wd)
    outs.append(dw)
  assert torch.equal(outs[0], outs[1])
  err = float((outs[0].float() - ref).abs().max() / ref.abs().max())
  assert err <= (1e-4 if dw_dtype == torch.float32 else 6e-3), err


@pytest.mark.parametrize('imgs,H,Wd,C,N,k,pad', [(4, 16, 16, 64, 128, 3, 1), (2, 28, 28, 128, 256, 3, 1), (3, 14, 10, 72, 64, 5, 2),
                                                 (2, 8, 8, 64, 64, 2, 0)])
def test_conv2d_backward_data_of_strided_convolutions_by_parity_classes(hip, imgs, H, Wd, C, N, k, pad):
  """pf_conv2d_bwd_data_strided (stride 2): four stride-1 launches of the implicit-GEMM kernel over dY, each walking a sub-grid of
  the flipped / transposed kernel in place and scattering its rows to one output-parity class, against autograd.  Every pixel of
  dX is written (the NaN fill must be gone); a second call gives the same bits."""
  g = torch.Generator(device='cuda').manual_seed(21 + k)
  stride = 2
  Ho, Wo = (H + 2 * pad - k) // stride + 1, (Wd + 2 * pad - k) // stride + 1
  x = _bf(torch.randn(imgs, H, Wd, C, device='cuda', generator=g))
  w = _bf(torch.randn(N, k, k, C, device='cuda', generator=g) * 0.05)
  dy = _bf(torch.randn(imgs, Ho, Wo, N, device='cuda', generator=g) * 0.1)
  xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
  F.conv2d(xt, w.float().permute(0, 3, 1, 2), stride=stride, padding=pad).backward(dy.float().permute(0, 3, 1, 2))
  ref = xt.grad.permute(0, 2, 3, 1)
  wb = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()            # [C][k][k][N]
  dx = torch.full((imgs, H, Wd, C), float('nan'), device='cuda').bfloat16()
  hip.conv2d_bwd_data_strided(dy, wb, dx, imgs, H, Wd, C, N, k, k, stride, pad, pad, Ho, Wo)
  torch.cuda.synchronize()
  assert torch.isfinite(dx.float()).all()
  _close(dx, _bf(ref), 'strided bwd-data')
  dx2 = torch.empty_like(dx)
  hip.conv2d_bwd_data_strided(dy, wb, dx2, imgs, H, Wd, C, N, k, k, stride, pad, pad, Ho, Wo)
  assert torch.equal(dx, dx2)
  # ... with the BN-backward sums of the BN in front of the convolution in the class launches' epilogues (round 6): the same dX
  # bits, the sums against pf_bn_bwd_stats over the stored dX
  M = imgs * H * Wd
  bnx = _bf(torch.randn(M, C, device='cuda', generator=g))
  ss = torch.stack([torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g) * 0.3])
  mi = torch.stack([torch.randn(C, device='cuda', generator=g) * 0.1, torch.rand(C, device='cuda', generator=g) + 0.5])
  G = hip.conv2d_bwd_data_strided_stats_groups(imgs, H, Wd, C, stride)
  assert G > 0 and G % (stride * stride) == 0
  partial = torch.full((G, 2, C), float('nan'), device='cuda')
  dx3 = torch.full_like(dx, float('nan'))
  hip.conv2d_bwd_data_strided(dy, wb, dx3, imgs, H, Wd, C, N, k, k, stride, pad, pad, Ho, Wo, partial=partial, bn_x=bnx,
                              bn_scale_shift=ss, bn_mean_invstd=mi, bn_act='Relu')
  assert torch.equal(dx3, dx) and not torch.isnan(partial).any()
  nblk = 16
  ref_partial = torch.empty(nblk * 2 * C, device='cuda')
  hip.bn_bwd_stats(dx.reshape(M, C), bnx, M, C, ss, mi, 'Relu', ref_partial, nblk)
  dgamma, dbeta = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
  hip.bn_bwd_finalize(ref_partial, nblk, C, dgamma, dbeta)
  dg2, db2 = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
  hip.bn_bwd_finalize(partial, G, C, dg2, db2)
  torch.testing.assert_close(db2, dbeta, rtol=1e-4, atol=1e-3)
  torch.testing.assert_close(dg2, dgamma, rtol=1e-4, atol=1e-3)

# ---- the MobileNet stem (pf_stem3.hip) --------------------------------------------------------------------------------
def _same(size, k=3, stride=2):
  out = -(-size // stride)
  total = max((out - 1) * stride + k - size, 0)
  return total // 2, total - total // 2, out


@pytest.mark.parametrize('N', [32, 16])
@pytest.mark.parametrize('imgs,H,Wd', [(3, 64, 64), (2, 224, 224), (1, 38, 96), (5, 63, 64), (130, 32, 64), (2, 17, 32)])
def test_conv_stem3_fwd_matches_torch(hip, imgs, H, Wd, N):
  """3x3 / stride 2, TensorFlow 'SAME' (asymmetric on even sizes: nothing in front, one row / column behind; symmetric on odd ones),
  3 -> 16 | 32 channels: against float32 torch on the explicitly padded image -- borders on all sides, a partial last strip of
  output rows, more (image, strip) items than workgroups, an asymmetric random kernel, repeated calls into a poisoned buffer."""
  (ph, ph1, Ho), (pw, pw1, Wo) = _same(H), _same(Wd)
  assert hip.conv_stem3_supported(H, Wd, 3, N, 3, 2, ph, pw, Ho, Wo) and not hip.conv_stem3_supported(H, Wd, 4, N, 3, 2, ph, pw, Ho, Wo)
  g = torch.Generator(device='cuda').manual_seed(H + Wd + imgs + N)
  x = _bf(torch.randn(imgs, H, Wd, 3, device='cuda', generator=g))
  w = _bf(torch.randn(N, 3, 3, 3, device='cuda', generator=g) * 0.2)
  xp = F.pad(x.float().permute(0, 3, 1, 2), (pw, pw1, ph, ph1))
  ref = _bf(F.conv2d(xp, w.float().permute(0, 3, 1, 2), stride=2).permute(0, 2, 3, 1))
  assert ref.shape == (imgs, Ho, Wo, N)
  for rep in range(2):
    y = torch.full((imgs, Ho, Wo, N), float('nan'), device='cuda', dtype=torch.bfloat16)
    hip.conv_stem3_fwd(x, w, y, imgs, H, Wd, N, ph, pw, Ho, Wo)
    _close(y, ref, 'stem3 %dx%dx%d N=%d call %d' % (imgs, H, Wd, N, rep))
  # one-hot probes: every (tap, channel) weight must meet exactly its input element
  x1 = torch.zeros(1, 32, 32, 3, device='cuda', dtype=torch.bfloat16)
  x1[0, 10, 13, 1] = 1.0
  x1[0, 31, 31, 2] = 1.0                                  # the last pixel: only reached through the padding behind the image
  y1 = torch.empty(1, 16, 16, N, device='cuda', dtype=torch.bfloat16)
  hip.conv_stem3_fwd(x1, w, y1, 1, 32, 32, N, 0, 0, 16, 16)
  r1 = _bf(F.conv2d(F.pad(x1.float().permute(0, 3, 1, 2), (0, 1, 0, 1)), w.float().permute(0, 3, 1, 2), stride=2).permute(0, 2, 3, 1))
  assert torch.equal(y1, r1)


@pytest.mark.parametrize('dw_dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('N', [32, 16])
@pytest.mark.parametrize('imgs,H,Wd', [(3, 64, 64), (2, 224, 224), (1, 38, 96), (5, 63, 64), (130, 32, 64)])
def test_conv_stem3_wrw_matches_autograd(hip, imgs, H, Wd, N, dw_dtype):
  """Backward-filter of the MobileNet stem against aten's float32 convolution_backward on the same bf16 operands (explicitly padded
  image): an odd number of output rows (a half-empty last strip), more items than workgroups, border taps, both gradient dtypes;
  bit-reproducible (fixed-order slab reduction)."""
  (ph, ph1, Ho), (pw, pw1, Wo) = _same(H), _same(Wd)
  S = hip.conv_stem3_wrw_slabs(imgs, H, Wd, N, ph, pw, Ho, Wo)
  assert S > 0
  g = torch.Generator(device='cuda').manual_seed(H * 3 + Wd + imgs + N)
  x = _bf(torch.randn(imgs, H, Wd, 3, device='cuda', generator=g))
  dy = _bf(torch.randn(imgs, Ho, Wo, N, device='cuda', generator=g) * 0.1)
  xp = F.pad(x.float().permute(0, 3, 1, 2), (pw, pw1, ph, ph1))
  ref = torch.ops.aten.convolution_backward(dy.float().permute(0, 3, 1, 2), xp, torch.zeros(N, 3, 3, 3, device='cuda'), None, [2, 2], [0, 0],
                                            [1, 1], False, [0, 0], 1, [False, True, False])[1].permute(0, 2, 3, 1)
  outs = []
  for rep in range(2):
    ws = torch.full(((S + 32) * N * 27,), float('nan'), device='cuda')
    dw = torch.full((N, 3, 3, 3), float('nan'), device='cuda', dtype=dw_dtype)
    hip.conv_stem3_wrw(dy, x, dw, ws, imgs, H, Wd, N, ph, pw, Ho, Wo)
    outs.append(dw.float())
  scale = float(ref.abs().max())
  torch.testing.assert_close(outs[0], ref, rtol=2e-2 if dw_dtype == torch.bfloat16 else 2e-3, atol=(8e-3 if dw_dtype == torch.bfloat16 else 2e-3) * scale)
  assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('tile', ['', '256x128'])
@pytest.mark.parametrize('imgs,H,C,N,stride', [(24, 28, 128, 128, 1), (16, 56, 64, 64, 1), (12, 28, 128, 128, 2), (37, 7, 512, 512, 1),
                                               (5, 9, 64, 72, 1)])
def test_conv2d_fwd_with_the_consumers_inference_bn_in_the_epilogue(hip, monkeypatch, tile, imgs, H, C, N, stride):
  """pf_conv2d_fwd_affine == pf_conv2d_fwd followed by pf_bn_act_quant_apply(quantize = 0), bit for bit; a tile override has no
  instantiation with the folded pass: plain launch + stand-alone pass in place."""
  if tile:
    monkeypatch.setenv('PF_IGEMM_TILE', tile)
  else:
    monkeypatch.delenv('PF_IGEMM_TILE', raising=False)
  hip.tuning_reload()
  try:
    g = torch.Generator(device='cuda').manual_seed(H + C + N)
    x = _bf(torch.randn(imgs, H, H, C, device='cuda', generator=g))
    w = _bf(torch.randn(N, 3, 3, C, device='cuda', generator=g) * 0.05)
    oss = torch.stack([torch.rand(N, device='cuda', generator=g) + 0.5, torch.randn(N, device='cuda', generator=g) * 0.5])
    y0 = _run(hip, x, w, stride, (1, 1))
    M = y0.numel() // N
    ref = torch.empty_like(y0)
    hip.bn_act_quant_apply(y0, ref, M, N, oss, 'Relu', None, 8, False)
    y1 = _run(hip, x, w, stride, (1, 1), out_scale_shift=oss, out_act='Relu')
    assert torch.equal(y1, ref), 'folded pass differs in %d elements' % int((y1 != ref).sum())
  finally:
    monkeypatch.delenv('PF_IGEMM_TILE', raising=False)
    hip.tuning_reload()


# ---- the window-staged kernel for 3x3 / stride 1, 64 -> 64 channels on 56 x 56 maps (pf_conv3x3_c64.hip) ---------------------------
def _h3_switch(hip, monkeypatch, on):
  monkeypatch.setenv('PF_CONV3X3_C64', '1' if on else '0')
  hip.tuning_reload()


@pytest.mark.parametrize('imgs', [1, 2, 19, 40])
def test_conv3x3_c64_window_kernel_equals_the_per_tap_kernel(hip, monkeypatch, imgs):
  """Stage-1 conv2 of ResNet-50 on its own kernel: two image rows per tile, the input window staged once for the nine taps, the kernel
  slice in registers.  Same k order and accumulation chain as the per-tap implicit GEMM -> the SAME BITS, for: one image (top and
  bottom padding rows in every other tile), more tiles than persistent workgroups (imgs >= 19), plain / residual + statistics /
  backward-data with BN-backward sums / output affine; against float32 torch too; a second launch into poisoned buffers."""
  H, C = 56, 64
  g = torch.Generator(device='cuda').manual_seed(imgs)
  x = _bf(torch.randn(imgs, H, H, C, device='cuda', generator=g))
  w = _bf(torch.randn(C, 3, 3, C, device='cuda', generator=g) * 0.05)        # asymmetric: a transposed window / tap order would show
  M = imgs * H * H
  r = _bf(torch.randn(M, C, device='cuda', generator=g))
  bnx = _bf(torch.randn(M, C, device='cuda', generator=g))
  ss = torch.stack([torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g) * 0.3])
  mi = torch.stack([torch.randn(C, device='cuda', generator=g) * 0.1, torch.rand(C, device='cuda', generator=g) + 0.5])
  geom = (imgs, H, H, C, C, 3, 3, 1, 1, 1, H, H)
  out = {}
  try:
    for on in (False, True):
      _h3_switch(hip, monkeypatch, on)
      G = hip.conv2d_stats_groups(M, C, geom=geom)
      for rep in range(2 if on else 1):
        y_plain = _run(hip, x, w, 1, (1, 1))
        p = torch.full((G, 4, C), float('nan'), device='cuda')
        y_res = _run(hip, x, w, 1, (1, 1), R=r, partial=p)
        pb = torch.full((G, 2, C), float('nan'), device='cuda')
        y_bwd = _run(hip, x, w, 1, (1, 1), partial=pb, bn_x=bnx, bn_scale_shift=ss, bn_mean_invstd=mi, bn_act='Relu')
        y_aff = _run(hip, x, w, 1, (1, 1), out_scale_shift=ss, out_act='Relu')
        cur = (y_plain, y_res, p, y_bwd, pb, y_aff, G)
        if on and rep == 1:
          assert all(torch.equal(a, b) for a, b in zip(cur[:6], out[True][:6])), 'two launches of the window kernel differ'
        out[on] = cur
  finally:
    monkeypatch.delenv('PF_CONV3X3_C64', raising=False)
    hip.tuning_reload()
  a, b = out[False], out[True]
  assert b[6] == min(imgs * 28, 512) and not torch.isnan(b[2]).any() and not torch.isnan(b[4]).any()
  for i, what in ((0, 'plain'), (1, 'residual'), (3, 'backward-data'), (5, 'output affine')):
    assert torch.equal(a[i], b[i]), '%s: window kernel and per-tap kernel differ in %d elements' % (what, int((a[i] != b[i]).sum()))
  _close(b[0], _bf(_ref(x, w, 1, (1, 1))), 'window kernel vs torch')
  # statistics: different partial groupings of the same stored values
  yf = b[1].float().reshape(M, C)
  torch.testing.assert_close(b[2][:, 0].sum(0), yf.sum(0), rtol=1e-4, atol=2e-2)
  torch.testing.assert_close(b[2][:, 1].sum(0), (yf * yf).sum(0), rtol=1e-4, atol=2e-2)
  assert torch.equal(b[2][:, 2].min(0).values, yf.min(0).values) and torch.equal(b[2][:, 3].max(0).values, yf.max(0).values)
  torch.testing.assert_close(b[4][:, 0].sum(0), a[4][:, 0].sum(0), rtol=1e-4, atol=1e-3)
  torch.testing.assert_close(b[4][:, 1].sum(0), a[4][:, 1].sum(0), rtol=1e-4, atol=1e-3)


def test_conv3x3_c64_window_kernel_one_hot_probes(hip):
  """Every (tap, channel) weight meets exactly its input element, at the image corners too (padding rows / columns are zeros)."""
  H, C = 56, 64
  g = torch.Generator(device='cuda').manual_seed(5)
  w = _bf(torch.randn(C, 3, 3, C, device='cuda', generator=g))
  for (hh, ww, cc) in ((0, 0, 3), (55, 55, 60), (0, 55, 17), (27, 1, 33), (28, 54, 0), (1, 0, 63)):
    x = torch.zeros(2, H, H, C, device='cuda', dtype=torch.bfloat16)
    x[1, hh, ww, cc] = 1.0
    y = _run(hip, x, w, 1, (1, 1))
    assert torch.equal(y, _bf(_ref(x, w, 1, (1, 1)))), (hh, ww, cc)


@pytest.mark.parametrize('imgs,dw_dtype', [(1, torch.float32), (2, torch.float32), (19, torch.float32), (40, torch.bfloat16)])
def test_wrw3x3_c64_window_kernel_matches_float64_and_the_shared_tile_kernel(hip, monkeypatch, imgs, dw_dtype):
  """Backward-filter of stage-1 conv2 on its own kernel (pf_wrw3x3_c64.hip: both operands staged once per two image rows for all nine
  taps, transposing LDS reads with per-lane pixel addresses, the [64][576] output in registers across the workgroup's tiles):
  against float64 torch on the same bf16 operands, against the shared-tile kernel (different summation order: float32 round-off),
  deterministic, padding taps exact (an ASYMMETRIC random kernel gradient: a transposed tap / channel order would show)."""
  H, C = 56, 64
  g = torch.Generator(device='cuda').manual_seed(100 + imgs)
  x = _bf(torch.randn(imgs, H, H, C, device='cuda', generator=g))
  dy = _bf(torch.randn(imgs, H, H, C, device='cuda', generator=g) * 0.1)
  M = imgs * H * H
  out = {}
  try:
    for on in (False, True):
      _h3_switch(hip, monkeypatch, on)
      S = hip.conv2d_wrw_splits(M, C, C, 9)
      assert S > 0
      ws = torch.full(((S + 32) * C * 9 * C,), float('nan'), device='cuda')
      dw = torch.full((C, 3, 3, C), float('nan'), device='cuda', dtype=dw_dtype)
      hip.conv2d_wrw(dy, x, dw, ws, imgs, H, H, C, C, 3, 3, 1, 1, 1, H, H)
      if on:
        ws2 = torch.full_like(ws, float('nan'))
        dw2 = torch.full_like(dw, float('nan'))
        hip.conv2d_wrw(dy, x, dw2, ws2, imgs, H, H, C, C, 3, 3, 1, 1, 1, H, H)
        assert torch.equal(dw, dw2)
      out[on] = dw
  finally:
    monkeypatch.delenv('PF_CONV3X3_C64', raising=False)
    hip.tuning_reload()
  ref = torch.empty(C, 3, 3, C, device='cuda', dtype=torch.float64)
  xp = F.pad(x, (0, 0, 1, 1, 1, 1))
  dyd = dy.reshape(M, C).double()
  for r in range(3):
    for s in range(3):
      ref[:, r, s, :] = dyd.t() @ xp[:, r:r + H, s:s + H, :].reshape(M, C).double()
  tol = 2e-5 if dw_dtype == torch.float32 else 6e-3
  for on in (False, True):
    err = float((out[on].double() - ref).abs().max() / ref.abs().max())
    assert err <= tol, ('window kernel' if on else 'shared-tile kernel', err)
