"""pf_convg.hip (general convolution / dense kernels: any shape and stride, float32 or bf16 storage, float32 accumulation) against
torch's float32 convolution and autograd, through the C ABI (pocketflow_amd/hip.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (imgs, H, W, C, N, R, stride, pad, bias)
SHAPES = [
    (4, 32, 32, 16, 16, 3, 1, 1, False),      # ResNet-20 stage 1
    (4, 32, 32, 16, 32, 3, 2, 1, False),      # its strided convolutions (backward-data with gaps)
    (3, 16, 16, 32, 64, 1, 2, 0, False),      # strided 1x1 projection
    (2, 15, 13, 3, 6, 5, 1, 0, True),         # LeNet: 5x5 VALID, 3 input channels (scalar loads), bias, odd sizes, N tail
    (2, 32, 32, 3, 16, 3, 1, 1, False),       # ResNet-20 first convolution
    (2, 30, 30, 3, 64, 7, 2, 3, False),       # a 7x7 / 2 stem outside the specialised kernel's sizes
    (5, 9, 9, 20, 10, 3, 1, 1, True),         # tails in every dimension (C % 4 == 0, N % 4 != 0)
    (8, 1, 1, 2048, 1001, 1, 1, 0, True),     # the dense layer 2048 -> 1001 as a 1x1 convolution
    (2, 14, 14, 256, 256, 3, 2, 1, False),    # a ResNet-50 strided 3x3, small
]


def _ref(x, w, b, stride, pad):
  return F.conv2d(x, w, b, stride=stride, padding=pad)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('imgs,H,W,C,N,R,stride,pad,bias', SHAPES)
def test_convg_forward_backward_data_backward_filter(dtype, imgs, H, W, C, N, R, stride, pad, bias):
  from pocketflow_amd import hip
  g = torch.Generator(device='cpu').manual_seed(1234 + H * 7 + C)
  x = torch.randn((imgs, C, H, W), generator=g).cuda().contiguous(memory_format=torch.channels_last)
  w = (torch.randn((N, C, R, R), generator=g) / float(np.sqrt(C * R * R))).cuda()
  b = torch.randn((N,), generator=g).cuda() if bias else None
  Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
  dy = torch.randn((imgs, N, Ho, Wo), generator=g).cuda().contiguous(memory_format=torch.channels_last)
  # what the kernel sees: values rounded to the storage dtype; the reference computes in float32 on the SAME values
  xs, ws, dys = x.to(dtype), w.to(dtype), dy.to(dtype)
  xr, wr, dyr = xs.float().requires_grad_(True), ws.float().requires_grad_(True), dys.float()
  yr = _ref(xr, wr, b, stride, pad)
  yr.backward(dyr)
  wk = ws.permute(0, 2, 3, 1).contiguous()
  y = torch.full((imgs, N, Ho, Wo), float('nan'), dtype=dtype, device='cuda').contiguous(memory_format=torch.channels_last)
  hip.convg_fwd(xs, wk, b, y, imgs, H, W, C, N, R, R, stride, pad, pad, Ho, Wo)
  dx = torch.full((imgs, C, H, W), float('nan'), dtype=dtype, device='cuda').contiguous(memory_format=torch.channels_last)
  hip.convg_bwd_data(dys, wk, dx, imgs, H, W, C, N, R, R, stride, pad, pad, Ho, Wo)
  splits = hip.convg_wrw_splits(imgs, C, N, R, R, Ho, Wo)
  slab = torch.full((splits * N * R * R * C,), float('nan'), dtype=torch.float32, device='cuda')
  dwk = torch.full((N, R, R, C), float('nan'), dtype=torch.float32, device='cuda')
  hip.convg_wrw(dys, xs, dwk, slab, imgs, H, W, C, N, R, R, stride, pad, pad, Ho, Wo)
  torch.cuda.synchronize()
  # float32 storage: summation order only (K up to 2304 terms: 1e-5 of the result's scale); bf16: one output rounding on top
  tol = 2e-5 if dtype == torch.float32 else 6e-3

  def close(got, ref, what):
    got, ref = got.float(), ref.float()
    assert torch.isfinite(got).all(), what
    scale = float(ref.abs().max()) + 1e-30
    err = float((got - ref).abs().max()) / scale
    assert err <= tol, '%s: max |err| / max |ref| = %.3e' % (what, err)
  close(y, yr.detach(), 'forward')
  close(dx, xr.grad, 'backward-data')
  # with a workspace, an output of few tiles (the dense layer) splits its contraction over slabs: same result up to summation order
  # (the split is a function of the shape -- hip.convg_small_splits -- and the workspace must hold splits * M * Nc floats)
  need = max(hip.convg_small_splits(imgs * Ho * Wo, N, R * R * C) * imgs * Ho * Wo * N,
             hip.convg_small_splits(imgs * H * W, C, R * R * N) * imgs * H * W * C, 1)
  ws = torch.full((need,), float('nan'), dtype=torch.float32, device='cuda')
  ys, dxs = torch.full_like(y, float('nan')), torch.full_like(dx, float('nan'))
  hip.convg_fwd(xs, wk, b, ys, imgs, H, W, C, N, R, R, stride, pad, pad, Ho, Wo, slab=ws)
  hip.convg_bwd_data(dys, wk, dxs, imgs, H, W, C, N, R, R, stride, pad, pad, Ho, Wo, slab=ws)
  close(ys, yr.detach(), 'forward (split contraction)')
  close(dxs, xr.grad, 'backward-data (split contraction)')
  close(dwk.permute(0, 3, 1, 2), wr.grad, 'backward-filter')
  # the summation order does not depend on how large the caller's workspace is (the layer executor's shared scratch grows
  # during a run): a 16x larger workspace gives the same bits
  big = torch.full((need * 16 + (1 << 22),), float('nan'), dtype=torch.float32, device='cuda')
  yb, dxb = torch.full_like(y, float('nan')), torch.full_like(dx, float('nan'))
  hip.convg_fwd(xs, wk, b, yb, imgs, H, W, C, N, R, R, stride, pad, pad, Ho, Wo, slab=big)
  hip.convg_bwd_data(dys, wk, dxb, imgs, H, W, C, N, R, R, stride, pad, pad, Ho, Wo, slab=big)
  assert torch.equal(ys, yb) and torch.equal(dxs, dxb)
  # deterministic: a second call gives the same bits
  y2 = torch.empty_like(y)
  hip.convg_fwd(xs, wk, b, y2, imgs, H, W, C, N, R, R, stride, pad, pad, Ho, Wo)
  dw2 = torch.empty_like(dwk)
  hip.convg_wrw(dys, xs, dw2, slab, imgs, H, W, C, N, R, R, stride, pad, pad, Ho, Wo)
  assert torch.equal(y, y2) and torch.equal(dwk, dw2)


def test_convg_wrw_writes_bf16_gradients(tmp_path):
  from pocketflow_amd import hip
  g = torch.Generator(device='cpu').manual_seed(7)
  imgs, H, W, C, N, R = 3, 8, 8, 16, 32, 3
  x = torch.randn((imgs, C, H, W), generator=g).cuda().contiguous(memory_format=torch.channels_last).bfloat16()
  dy = torch.randn((imgs, N, H, W), generator=g).cuda().contiguous(memory_format=torch.channels_last).bfloat16()
  slab = torch.empty((hip.convg_wrw_splits(imgs, C, N, R, R, H, W) * N * R * R * C,), dtype=torch.float32, device='cuda')
  d32 = torch.empty((N, R, R, C), dtype=torch.float32, device='cuda')
  d16 = torch.empty((N, R, R, C), dtype=torch.bfloat16, device='cuda')
  hip.convg_wrw(dy, x, d32, slab, imgs, H, W, C, N, R, R, 1, 1, 1, H, W)
  hip.convg_wrw(dy, x, d16, slab, imgs, H, W, C, N, R, R, 1, 1, 1, H, W)
  assert torch.equal(d32.bfloat16(), d16)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_layers_route_odd_shapes_to_the_general_kernels(tmp_path, dtype, monkeypatch):
  """graph.Conv2D / graph.Dense with shapes the MFMA kernels do not take: forward and gradients through the layer executor
  equal torch's own convolution (PF_OWN_CONV_GENERIC=0 path) on the same values."""
  import pocketflow_amd.graph as G
  from pocketflow_amd.graph import Graph, Conv2D, Dense
  cd = torch.float32 if dtype == 'float32' else torch.bfloat16
  outs = []
  for own in (True, False):
    monkeypatch.setattr(G, 'OWN_CONV_GENERIC', own)
    g = Graph('model', 'cuda', cd)
    with g.as_default():
      conv = Conv2D(g, 'c1', 3, 16, 3, 2, 'SAME', use_bias=True)
      dense = Dense(g, 'fc', 16 * 8 * 8, 10)
    g.finalize(seed=3)
    x = torch.randn((4, 3, 16, 16), generator=torch.Generator().manual_seed(5)).cuda().to(cd).contiguous(memory_format=torch.channels_last)
    with g.as_default():
      y = conv(x)
      z = dense(y.permute(0, 2, 3, 1).reshape(4, -1))
    loss = (z.float() ** 2).sum()
    loss.backward()
    outs.append((z.detach().float(), g.store.w_grad.detach().float().clone(), g.store.o_grad.detach().float().clone()))
  tol = 1e-4 if dtype == 'float32' else 3e-2
  for a, b in zip(outs[0], outs[1]):
    assert float((a - b).abs().max()) <= tol * (float(b.abs().max()) + 1e-6)


@pytest.mark.parametrize('imgs,H,W,C,N,R,stride,pad', [(4, 32, 32, 16, 16, 3, 1, 1), (4, 32, 32, 16, 32, 3, 2, 1), (3, 16, 16, 32, 32, 3, 1, 1),
                                                        (2, 16, 16, 32, 64, 3, 2, 1), (2, 9, 7, 48, 24, 5, 1, 2)])
def test_few_channel_convolutions_as_im2col_plus_1x1_kernels(imgs, H, W, C, N, R, stride, pad):
  """graph._ConvIm2col (pf_im2col.hip + the fused 1x1 kernels): forward, backward-data (col2im gather) and backward-filter of the
  ResNet-20 convolutions against torch's float32 convolution on the same bf16 values."""
  import pocketflow_amd.graph as G
  from pocketflow_amd.graph import Graph, Conv2D
  g = Graph('model', 'cuda', torch.bfloat16)
  with g.as_default():
    conv = Conv2D(g, 'c', C, N, R, stride, padding=pad, use_bias=False)
  g.finalize(seed=11)
  gen = torch.Generator().manual_seed(3 + C + H)
  x = torch.randn((imgs, C, H, W), generator=gen).cuda().bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
  assert G.im2col_conv_ok(x, conv, (pad, pad))
  with g.as_default():
    y = conv(x)
  Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
  dy = torch.randn((imgs, N, Ho, Wo), generator=gen).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
  y.backward(dy)
  xr = x.detach().float().requires_grad_(True)
  wr = conv.kernel.tensor.detach().float().requires_grad_(True)
  yr = F.conv2d(xr, wr, None, stride=stride, padding=pad)
  yr.backward(dy.float())

  def close(got, ref, what, tol=1.2e-2):
    got, ref = got.float(), ref.float()
    assert torch.isfinite(got).all(), what
    err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
    assert err <= tol, '%s: %.3e' % (what, err)
  close(y.detach(), yr.detach(), 'forward')
  close(x.grad, xr.grad, 'backward-data')
  v = conv.kernel
  gw = g.store.w_grad[v.offset:v.offset + v.numel].view(N, R, R, C).permute(0, 3, 1, 2)
  close(gw, wr.grad, 'backward-filter')
