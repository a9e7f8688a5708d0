"""The autograd plumbing of the fused path (graph.LazyAct, _BnLazy, _FusedConv1x1: prologue hand-off,
statistics hand-off from the convolution epilogue to the next BN, residual in the epilogue, shortcut
gradient folded into the BN backward, BN-backward statistics from the backward-data kernel, gradients
written straight into the flat buffers) checked EXACTLY on the CPU: the HIP entry points are replaced by
float32 torch emulations of their documented semantics (include/pocketflow_hip.h), and a small bottleneck
ResNet is run with fusion on and off -- both must give the same outputs, gradients and BN state to float32
round-off.  The GPU tests can only compare the bf16 kernels statistically (tests/test_conv_gpu.py); this
test pins the LOGIC around them."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _rows(t, C):
  """[rows][C] view of a logical-NCHW / physical-NHWC tensor (or a flat [rows][C] one)."""
  if t.dim() == 4:
    return t.permute(0, 2, 3, 1).reshape(-1, C)
  return t.reshape(-1, C)


def _act(y, act):
  if act in ('Relu', 'relu'):
    return torch.relu(y)
  if act in ('Relu6', 'relu6'):
    return torch.clamp(y, 0, 6)
  return y


def _mask(u, act):
  if act in ('Relu', 'relu'):
    return (u > 0).float()
  if act in ('Relu6', 'relu6'):
    return ((u > 0) & (u < 6)).float()
  return torch.ones_like(u)


class FakeHip(object):
  """float32 torch emulation of the entry points graph.py calls; min/max slots hold two float32 bit patterns."""

  def __init__(self):
    self.calls = {}

  def _n(self, name):
    self.calls[name] = self.calls.get(name, 0) + 1

  # -- slots ----------------------------------------------------------------------------------------
  def minmax_slots_init(self, slots):
    slots.view(-1).copy_(torch.tensor([float('inf'), float('-inf')] * (slots.numel() // 2)).view(torch.int32))

  @staticmethod
  def _slot_get(slot):
    v = slot.view(torch.float32)
    return float(v[0]), float(v[1])

  @staticmethod
  def _quant(y, slot, bits):
    mn, mx = FakeHip._slot_get(slot)
    alpha, beta = (mx - mn) + 1e-10, mn
    k = float(2 ** bits - 1)
    return alpha * (torch.round((y - beta) / alpha * k) / k) + beta

  # -- BN forward --------------------------------------------------------------------------------------
  def bn_stats(self, x, rows, C, partial, nblk):
    self._n('bn_stats')
    xr = _rows(x, C).float()
    p = partial[:nblk * 4 * C].view(nblk, 4, C)
    p[:, 0:2] = 0
    p[:, 2] = float('inf')
    p[:, 3] = float('-inf')
    d = xr - xr[0:1]
    p[0, 0], p[0, 1], p[0, 2], p[0, 3] = d.sum(0), (d * d).sum(0), xr.min(0).values, xr.max(0).values

  def bn_finalize(self, partial, nblk, rows, C, piv, gamma, beta, mm, mv, momentum, eps, training, act, ss, mi, slot):
    self._n('bn_finalize')
    p = partial.reshape(-1)[:nblk * 4 * C].view(nblk, 4, C).double()
    pivot = _rows(piv, C)[0].double() if piv.numel() >= C and piv.dim() != 1 else piv.reshape(-1)[:C].double()
    s, q = p[:, 0].sum(0), p[:, 1].sum(0)
    mn, mx = p[:, 2].min(0).values.float(), p[:, 3].max(0).values.float()
    if training:
      m1 = s / rows
      var = (q / rows - m1 * m1).clamp_min(0)
      mean = (pivot + m1).float()
      unbiased = (var * (rows / max(rows - 1, 1))).float()
      mm.sub_((mm - mean) * (1 - momentum))
      mv.sub_((mv - unbiased) * (1 - momentum))
      var = var.float()
    else:
      mean, var = mm.clone(), mv.clone()
    invstd = 1.0 / torch.sqrt(var + eps)
    sc = gamma.detach() * invstd
    ss[0], ss[1] = sc, beta.detach() - mean * sc
    mi[0], mi[1] = mean, invstd
    if slot is not None:
      a, b = sc * mn + ss[1], sc * mx + ss[1]
      ymin, ymax = _act(torch.minimum(a, b), act).min(), _act(torch.maximum(a, b), act).max()
      cur = slot.view(torch.float32)
      cur[0], cur[1] = min(float(cur[0]), float(ymin)), max(float(cur[1]), float(ymax))

  def bn_eval_scale_shift(self, gamma, beta, mm, mv, eps, ss):
    sc = gamma.detach() / torch.sqrt(mv + eps)
    ss[0], ss[1] = sc, beta.detach() - mm * sc

  def _q_of(self, xr, ss, act, slot, bits, quantize):
    y = _act(xr * ss[0] + ss[1], act)
    return self._quant(y, slot, bits) if quantize else y

  def bn_act_quant_apply(self, x, q, rows, C, ss, act, slot, bits, quantize):
    self._n('bn_apply')
    _rows(q, C).copy_(self._q_of(_rows(x, C).float(), ss, act, slot, bits, quantize))

  # -- BN backward -------------------------------------------------------------------------------------------
  def bn_bwd_stats(self, dq, x, rows, C, ss, mi, act, partial, nblk):
    self._n('bn_bwd_stats')
    xr, g = _rows(x, C).float(), _rows(dq, C).float()
    dy = g * _mask(xr * ss[0] + ss[1], act)
    p = partial[:nblk * 2 * C].view(nblk, 2, C)
    p.zero_()
    p[0, 0], p[0, 1] = dy.sum(0), (dy * (xr - mi[0]) * mi[1]).sum(0)

  def bn_bwd_finalize(self, partial, nblk, C, dgamma, dbeta):
    p = partial.reshape(-1)[:nblk * 2 * C].view(nblk, 2, C)
    dbeta.copy_(p[:, 0].sum(0))
    dgamma.copy_(p[:, 1].sum(0))

  def bn_bwd_apply(self, dq, x, dx, rows, C, ss, mi, dgamma, dbeta, act, addend=None):
    self._n('bn_bwd_apply_add' if addend is not None else 'bn_bwd_apply')
    xr, g = _rows(x, C).float(), _rows(dq, C).float()
    dy = g * _mask(xr * ss[0] + ss[1], act)
    out = ss[0] * (dy - dbeta / rows - (xr - mi[0]) * mi[1] * dgamma / rows)
    if addend is not None:
      out = out + _rows(addend, C).float()
    _rows(dx, C).copy_(out)

  # -- fused 1x1 convolutions -------------------------------------------------------------------------------------
  def conv1x1_stats_groups(self, M, N):
    return 3

  def conv1x1_wrw_splits(self, M, N, K):
    return 2

  @staticmethod
  def _gather(x, K, geom):
    if geom is None:
      return _rows(x, K).float()
    ho, wo, h, w, s = geom
    return x.permute(0, 2, 3, 1)[:, ::s, ::s, :][:, :ho, :wo, :].reshape(-1, K).float()

  def conv1x1_fwd(self, X, W, Y, M, N, K, R=None, scale_shift=None, act=None, slot=None, bits=8, partial=None,
                  geom=None, ymap=False):
    self._n('conv1x1_fwd' if scale_shift is not None else 'conv1x1_plain')
    if ymap:                                   # backward-data of a strided conv: scatter rows
      ho, wo, h, w, s = geom
      out = _rows(X, K).float() @ W.float().t()
      Y.permute(0, 2, 3, 1)[:, ::s, ::s, :][:, :ho, :wo, :] = out.view(Y.shape[0], ho, wo, N)
      return
    xr = self._gather(X, K, geom)
    if scale_shift is not None:
      xr = self._q_of(xr, scale_shift, act, slot, bits, slot is not None)
    y = xr @ W.float().t()
    if R is not None:
      y = y + _rows(R, N).float()
    _rows(Y, N).copy_(y)
    if partial is not None:
      partial[:, 0:2] = 0
      partial[:, 2] = float('inf')
      partial[:, 3] = float('-inf')
      partial[1, 0], partial[1, 1], partial[1, 2], partial[1, 3] = y.sum(0), (y * y).sum(0), y.min(0).values, y.max(0).values

  def conv1x1_bwd_data_bnstats(self, dY, Wt, dQ, bn_x, bn_ss, bn_mi, bn_act, partial, M, N, K):
    self._n('conv1x1_bwd_data_bnstats')
    dq = _rows(dY, N).float() @ Wt.float().t()
    _rows(dQ, K).copy_(dq)
    xr = _rows(bn_x, K).float()
    dy = dq * _mask(xr * bn_ss[0] + bn_ss[1], bn_act)
    partial.zero_()
    partial[2, 0], partial[2, 1] = dy.sum(0), (dy * (xr - bn_mi[0]) * bn_mi[1]).sum(0)

  def conv1x1_wrw(self, dY, X, dW, workspace, M, N, K, scale_shift=None, act=None, slot=None, bits=8, geom=None):
    self._n('conv1x1_wrw')
    xr = self._gather(X, K, geom)
    if scale_shift is not None:
      xr = self._q_of(xr, scale_shift, act, slot, bits, slot is not None)
    dW.copy_(_rows(dY, N).float().t() @ xr)


def _build(fuse, fake, act_bits):
  from pocketflow_amd import graph as G
  from pocketflow_amd.utils.external import resnet_model as R
  g = G.Graph('model', 'cpu', torch.float32)
  g.fuse_conv1x1 = fuse
  net = R.Model(50, True, 7, 8, 3, 1, None, None, [2, 2], [1, 2], data_format='channels_last', graph=g)
  g.finalize(seed=3, requires_grad=True)
  for op in g.activation_ops:
    op.bits = act_bits
  return g, net


@pytest.mark.parametrize('act_bits', [None, 6])
def test_fused_plumbing_is_exact_on_cpu(monkeypatch, act_bits):
  from pocketflow_amd import graph as G
  results = {}
  for fuse in (False, True):
    fake = FakeHip()
    monkeypatch.setattr(G, 'hip', fake)
    monkeypatch.setattr(G, 'fusable_tensor', lambda t: True)
    g, net = _build(fuse, fake, act_bits)
    torch.manual_seed(0)
    x = torch.randn(4, 3, 12, 12).contiguous(memory_format=torch.channels_last)
    wts = torch.randn(4, 7)
    g.begin_step = lambda: None
    fake.minmax_slots_init(g.act_slots)
    with g.as_default():
      logits = net(x, True)
    loss = (logits * wts).sum()
    loss.backward()
    st = g.store
    results[fuse] = dict(logits=logits.detach().clone(), w_grad=st.w_grad.clone(), o_grad=st.o_grad.clone(),
                         state=st.state.clone(), calls=dict(fake.calls))
  a, b = results[False], results[True]
  # the fused run really took the fused route: no materialised bn1/bn3, statistics from the epilogues, ...
  assert b['calls'].get('conv1x1_fwd', 0) == 2 * 4 + 2 and a['calls'].get('conv1x1_fwd', 0) == 0
  assert b['calls']['bn_apply'] < a['calls']['bn_apply'] and b['calls']['bn_stats'] < a['calls']['bn_stats']
  assert b['calls'].get('conv1x1_bwd_data_bnstats', 0) >= 4 and b['calls'].get('bn_bwd_apply_add', 0) == 2
  for k in ('logits', 'w_grad', 'o_grad', 'state'):
    ref, got = a[k], b[k]
    err = float((ref - got).abs().max() / (ref.abs().max() + 1e-12))
    # exact (float32 round-off) without quantisers; a 6-bit quantiser flips a few elements on that round-off
    tol = 2e-5 if act_bits is None else 0.25
    assert err <= tol, (k, err)
  if act_bits is not None:
    # ... and the gradients still point the same way
    cos = float(torch.dot(a['w_grad'], b['w_grad']) / (a['w_grad'].norm() * b['w_grad'].norm()))
    assert cos > 0.995, cos


def test_mobilenet_pointwise_fusion_is_exact_on_cpu(monkeypatch):
  """MobileNet-v1: the depthwise BN+ReLU6 is applied inside the pointwise 1x1 convolution, which leaves the
  statistics for its own BN; same exactness check as above (no quantisers)."""
  from pocketflow_amd import graph as G
  from pocketflow_amd.utils.external.mobilenet_v1 import MobilenetV1
  results = {}
  for fuse in (False, True):
    fake = FakeHip()
    monkeypatch.setattr(G, 'hip', fake)
    monkeypatch.setattr(G, 'fusable_tensor', lambda t: True)
    g = G.Graph('model', 'cpu', torch.float32)
    g.fuse_conv1x1 = fuse
    net = MobilenetV1(g, num_classes=16, depth_multiplier=0.25, dropout_keep_prob=1.0)   # 16 classes: fusable logits conv
    g.finalize(seed=5, requires_grad=True)
    torch.manual_seed(1)
    x = torch.randn(8, 3, 64, 64).contiguous(memory_format=torch.channels_last)
    wts = torch.randn(8, 16)
    fake.minmax_slots_init(g.act_slots)
    with g.as_default():
      logits = net(x, True)
    (logits * wts).sum().backward()
    st = g.store
    results[fuse] = dict(logits=logits.detach().clone(), w_grad=st.w_grad.clone(), o_grad=st.o_grad.clone(),
                         state=st.state.clone(), calls=dict(fake.calls))
  a, b = results[False], results[True]
  assert b['calls'].get('conv1x1_fwd', 0) == 13 and a['calls'].get('conv1x1_fwd', 0) == 0     # 13 pointwise convs
  assert b['calls']['bn_stats'] == a['calls']['bn_stats'] - 13                                # their BNs use the epilogue
  for k in ('logits', 'w_grad', 'o_grad', 'state'):
    err = float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-12))
    assert err <= 5e-4, (k, err)      # 27 BN layers deep; the fused statistics are un-pivoted sums


@pytest.mark.parametrize('dataset,size,ncls,shape', [('cifar_10', 20, 10, (32, 32, 3)), ('ilsvrc_12', 50, 11, (64, 64, 3)),
                                                     ('ilsvrc_12', 18, 7, (64, 64, 3))])
def test_product_resnet_matches_the_reference_network_code(monkeypatch, dataset, size, ncls, shape):
  """The PRODUCT's ResNet (pocketflow_amd/utils/external/resnet_model.py on the layer executor, HIP entry
  points emulated) loaded with reference-named variables reproduces the logits obtained by executing the
  reference's own utils/external/resnet_model.py (tests/golden fixtures): same variable names (load_numpy is
  strict), same creation order, same padding / projection / BN rules -- fused and unfused."""
  import os
  from oracle import learner_oracle as LO
  from pocketflow_amd import graph as G
  from pocketflow_amd.utils.external import resnet_model as R
  with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_arrays.npz')) as z:
    ref = {m: z['net/resnet_%s_%d/%s' % (dataset, size, m)] for m in ('train', 'eval')}
  vals, images = LO.net_fixture_recipe('resnet', dataset, size, ncls, shape)
  cfg = LO.resnet_cfg(dataset, size)
  for fuse in (False, True):
    fake = FakeHip()
    monkeypatch.setattr(G, 'hip', fake)
    monkeypatch.setattr(G, 'fusable_tensor', lambda t: True)
    g = G.Graph('model', 'cpu', torch.float32)
    g.fuse_conv1x1 = fuse
    net = R.Model(size, cfg['bottleneck'], ncls, cfg['num_filters'], cfg['kernel_size'], cfg['conv_stride'],
                  cfg['first_pool_size'], cfg['first_pool_stride'], cfg['block_sizes'], cfg['block_strides'],
                  data_format='channels_last', graph=g)
    g.finalize(seed=1, requires_grad=True)
    g.store.load_numpy(vals, strict=True)                 # every product variable exists under the reference name
    assert sorted(v.name for v in g.store.vars) == sorted(vals.keys())
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_host.json')) as f:
      order = json.load(f)['net_matmul_order']['resnet_%s_%d' % (dataset, size)]
    assert [op.var.name for op in g.matmul_ops] == order  # TF's creation order of the matmul kernels
    x = torch.from_numpy(images).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    for mode in ('eval', 'train'):
      fake.minmax_slots_init(g.act_slots)
      with g.as_default():
        if mode == 'eval':
          with torch.no_grad():
            logits = net(x, False)
        else:
          logits = net(x, True)
      got = logits.detach().numpy()
      assert np.max(np.abs(got - ref[mode])) <= 5e-4 * max(1.0, float(np.max(np.abs(ref[mode])))), (fuse, mode)
      g.store.load_numpy(vals, strict=True)               # undo the moving-average update of the training pass


def test_product_lenet_matches_the_reference_network_code():
  """Same for LeNet (nets/lenet_at_cifar10.py:forward_fn executed by the golden generator)."""
  import os
  from oracle import learner_oracle as LO
  from pocketflow_amd import graph as G
  from pocketflow_amd.nets.lenet_at_cifar10 import _LeNet
  with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_arrays.npz')) as z:
    ref = z['net/lenet_cifar_10_0/eval']
  vals, images = LO.net_fixture_recipe('lenet', 'cifar_10', 0, 10, (32, 32, 3))
  g = G.Graph('model', 'cpu', torch.float32)
  net = _LeNet(g, (32, 32, 3), 10)
  g.finalize(seed=1, requires_grad=False)
  g.store.load_numpy(vals, strict=True)
  assert sorted(v.name for v in g.store.vars) == sorted(vals.keys())
  with torch.no_grad(), g.as_default():
    got = net(torch.from_numpy(images).permute(0, 3, 1, 2), False).numpy()
  assert np.max(np.abs(got - ref)) <= 1e-5


@pytest.mark.parametrize('dm,shape,ncls', [(50, (64, 64, 3), 16), (100, (96, 96, 3), 8)])
def test_product_mobilenet_matches_the_reference_network_code(monkeypatch, dm, shape, ncls):
  """The product's MobileNet-v1 (slim naming, SAME padding, ReLU6, BN 0.9997/1e-3, biased logits conv) against the
  logits of the reference's own slim module, fused (depthwise BN inside the pointwise convolutions) and unfused."""
  import os
  from oracle import learner_oracle as LO
  from pocketflow_amd import graph as G
  from pocketflow_amd.utils.external.mobilenet_v1 import MobilenetV1
  here = os.path.dirname(os.path.abspath(__file__))
  with np.load(os.path.join(here, 'golden', 'reference_arrays.npz')) as z:
    ref = {m: z['net/mobilenet_v1_%d/%s' % (dm, m)] for m in ('train', 'eval')}
  with open(os.path.join(here, 'golden', 'reference_host.json')) as f:
    order = json.load(f)['net_matmul_order']['mobilenet_v1_%d' % dm]
  vals, images = LO.net_fixture_recipe('mobilenet_v1', 'ilsvrc_12', dm, ncls, shape)
  for fuse in (False, True):
    fake = FakeHip()
    monkeypatch.setattr(G, 'hip', fake)
    monkeypatch.setattr(G, 'fusable_tensor', lambda t: True)
    g = G.Graph('model', 'cpu', torch.float32)
    g.fuse_conv1x1 = fuse
    net = MobilenetV1(g, num_classes=ncls, depth_multiplier=dm / 100.0, dropout_keep_prob=1.0)
    g.finalize(seed=1, requires_grad=True)
    g.store.load_numpy(vals, strict=True)
    assert sorted(v.name for v in g.store.vars) == sorted(vals.keys())
    assert [op.var.name for op in g.matmul_ops] == order
    x = torch.from_numpy(images).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    for mode in ('eval', 'train'):
      fake.minmax_slots_init(g.act_slots)
      with g.as_default():
        if mode == 'eval':
          with torch.no_grad():
            logits = net(x, False)
        else:
          logits = net(x, True)
      got = logits.detach().numpy()
      assert np.max(np.abs(got - ref[mode])) <= 1e-3 * max(1.0, float(np.max(np.abs(ref[mode])))), (fuse, mode)
      g.store.load_numpy(vals, strict=True)
