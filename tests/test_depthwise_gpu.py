"""Depthwise 3x3 convolution kernels (pf_depthwise.hip: forward + BN statistics, backward-data, backward-filter) against
float32 torch references built from the SAME inputs -- float32 storage to summation-order accuracy, bf16 storage to one
bf16 ulp -- with TensorFlow's 'SAME' padding (front pad floor(total / 2): asymmetric for stride 2 on even sizes), odd and
even sizes, every channel count of MobileNet-v1 (32 ... 1024), strip tails (Wo % 4 != 0)."""
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


def _same(size, k, stride):
  out = -(-size // stride)
  total = max((out - 1) * stride + k - size, 0)
  return total // 2, total - total // 2, out


def _ref(x, w, stride):
  """float32 reference on logical NCHW tensors, TF 'SAME' padding"""
  ph0, ph1, _ = _same(x.shape[2], 3, stride)
  pw0, pw1, _ = _same(x.shape[3], 3, stride)
  return F.conv2d(F.pad(x, (pw0, pw1, ph0, ph1)), w, stride=stride, groups=x.shape[1])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,H,W,C,stride', [(3, 15, 13, 32, 1), (2, 16, 20, 64, 2), (5, 7, 7, 1024, 1), (2, 14, 14, 512, 2),
                                            (4, 28, 28, 256, 1), (2, 112, 112, 32, 1), (2, 112, 112, 64, 2), (1, 9, 6, 128, 2)])
def test_depthwise_fwd_bwd_wrw_match_torch(hip, monkeypatch, dtype, B, H, W, C, stride):
  assert hip.depthwise_supported(C, 3, stride)
  g = torch.Generator(device='cuda').manual_seed(B + H + W + C + stride)
  x = torch.randn(B, C, H, W, device='cuda', generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
  w = (torch.randn(C, 1, 3, 3, device='cuda', generator=g) * 0.3).to(dtype)
  ph, _, Ho = _same(H, 3, stride)
  pw, _, Wo = _same(W, 3, stride)
  xf = x.float().requires_grad_(True)
  wf = w.float().requires_grad_(True)
  ref = _ref(xf, wf, stride)
  # forward + statistics
  y = torch.full((B, C, Ho, Wo), float('nan'), device='cuda', dtype=dtype).contiguous(memory_format=torch.channels_last)
  G = hip.depthwise_groups(B, Ho, Wo, C)
  partial = torch.full((G, 4, C), float('nan'), device='cuda')
  hip.depthwise_fwd(x, w.reshape(C, 3, 3), y, B, H, W, C, 3, stride, ph, pw, Ho, Wo, partial=partial)
  tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=1e-3)
  torch.testing.assert_close(y.float(), ref.detach(), **tol)
  yr = y.float().permute(0, 2, 3, 1).reshape(-1, C)
  assert not torch.isnan(partial).any()
  torch.testing.assert_close(partial[:, 0].sum(0), yr.sum(0), rtol=1e-4, atol=1e-2)
  torch.testing.assert_close(partial[:, 1].sum(0), (yr * yr).sum(0), rtol=1e-4, atol=1e-2)
  assert torch.equal(partial[:, 2].min(0).values, yr.min(0).values) and torch.equal(partial[:, 3].max(0).values, yr.max(0).values)
  # backward
  dy = torch.randn(B, C, Ho, Wo, device='cuda', generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
  ref.backward(dy.float())
  dx = torch.full_like(x, float('nan'))
  hip.depthwise_bwd_data(dy, w.reshape(C, 3, 3), dx, B, H, W, C, 3, stride, ph, pw, Ho, Wo)
  torch.testing.assert_close(dx.float(), xf.grad, **(dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=5e-3)))
  for dw_dtype in (torch.float32, dtype):
    dw = torch.full((C, 3, 3), float('nan'), device='cuda', dtype=dw_dtype)
    slabs = torch.full(((G + 32) * C * 9,), float('nan'), device='cuda')
    hip.depthwise_wrw(dy, x, dw, slabs, B, H, W, C, 3, stride, ph, pw, Ho, Wo)
    scale = float(wf.grad.abs().max())
    err = float((dw.float() - wf.grad.reshape(C, 3, 3)).abs().max())
    assert err <= (1e-4 if dw_dtype == torch.float32 else 2 ** -7) * scale + 1e-5, (err, scale)
  # determinism: fixed-order slab reduction
  dw2 = torch.empty((C, 3, 3), device='cuda', dtype=torch.float32)
  dw3 = torch.empty_like(dw2)
  hip.depthwise_wrw(dy, x, dw2, slabs, B, H, W, C, 3, stride, ph, pw, Ho, Wo)
  hip.depthwise_wrw(dy, x, dw3, slabs, B, H, W, C, 3, stride, ph, pw, Ho, Wo)
  assert torch.equal(dw2, dw3)


def test_depthwise_layer_through_the_executor_matches_torch(hip):
  """graph.DepthwiseConv2D on the in-tree kernels: forward value, statistics left for the BatchNorm, autograd gradients
  (dx through pf_depthwise_bwd_data, dW written straight into the flat gradient buffer)."""
  from pocketflow_amd import graph as G
  gr = G.Graph('model', 'cuda', torch.float32)
  layer = G.DepthwiseConv2D(gr, 'dw', 64, 3, 2)
  gr.finalize(seed=3)
  x = torch.randn(4, 64, 20, 20, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
  with gr.as_default():
    y = layer(x, want_stats=True)
  assert y.shape == (4, 64, 10, 10) and hasattr(y, '_pf_stats')
  w = layer.kernel.tensor
  xr = x.detach().clone().requires_grad_(True)
  wr = w.detach().clone().requires_grad_(True)
  ref = _ref(xr, wr, 2)
  torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)
  gy = torch.randn_like(ref)
  y.backward(gy)
  ref.backward(gy)
  torch.testing.assert_close(x.grad, xr.grad, rtol=1e-5, atol=1e-5)
  st = gr.store
  got = st.w_grad[layer.kernel.offset:layer.kernel.offset + layer.kernel.numel].view(64, 1, 3, 3)
  torch.testing.assert_close(got, wr.grad, rtol=1e-4, atol=1e-4)
