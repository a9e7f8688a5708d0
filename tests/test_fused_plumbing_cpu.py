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


from fake_hip import FakeHip  # noqa: E402  (tests/fake_hip.py: float32 torch emulation of the HIP entry points)


def _build(fuse, fake, act_bits, filters=8):
  from pocketflow_amd import graph as G
  from pocketflow_amd.utils.external import resnet_model as R
  g = G.Graph('model', 'cpu', torch.float32)
  g.fuse_conv1x1 = fuse
  net = R.Model(50, True, 7, filters, 3, 1, None, None, [2, 2], [1, 2], data_format='channels_last', graph=g)
  g.finalize(seed=3, requires_grad=True)
  for op in g.activation_ops:
    op.bits = act_bits
  return g, net


@pytest.mark.parametrize('act_bits,filters', [(None, 8), (6, 8), (None, 64)])
def test_fused_plumbing_is_exact_on_cpu(monkeypatch, act_bits, filters):
  from pocketflow_amd import graph as G
  results = {}
  for fuse in (False, True):
    fake = FakeHip()
    monkeypatch.setattr(G, 'hip', fake)
    monkeypatch.setattr(G, 'fusable_tensor', lambda t: True)
    g, net = _build(fuse, fake, act_bits, filters)
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
  if filters == 64 and G.OWN_CONV2D:
    # C % 64 == 0: the four 3x3 convolutions run on the implicit-GEMM entry point (pf_conv2d_fwd) and leave bn3's
    # statistics; the three stride-1 ones also run backward-data there, with bn2's BN-backward sums in the epilogue
    assert b['calls'].get('conv2d_fwd', 0) == 4 and b['calls'].get('conv2d_bwd_data_bnstats', 0) == 3, b['calls']
    assert a['calls'].get('conv2d_fwd', 0) == 0
  for k in ('logits', 'w_grad', 'o_grad', 'state'):
    ref, got = a[k], b[k]
    err = float((ref - got).abs().max() / (ref.abs().max() + 1e-12))
    # exact (float32 round-off) without quantisers; a 6-bit quantiser flips a few elements on that round-off
    tol = 2e-5 if act_bits is None else 0.25
    if filters == 64:
      # wide layers: the epilogue statistics are un-pivoted float32 sums (sum x, sum x^2), whose cancellation error the BN
      # backward amplifies into the kernel gradients (3.3e-3 here, with or without the 3x3 path: PF_OWN_CONV2D=0 gives the
      # same figure); immaterial next to bf16 rounding, which is the only mode the fused path runs in on the GPU
      tol = 1e-2
    assert err <= tol, (k, err)
  if act_bits is not None:
    # ... and the gradients still point the same way
    cos = float(torch.dot(a['w_grad'], b['w_grad']) / (a['w_grad'].norm() * b['w_grad'].norm()))
    assert cos > 0.995, cos


@pytest.mark.parametrize('filters', [8, 64])
def test_inference_bn_folded_into_the_producing_convolution_is_exact_on_cpu(monkeypatch, filters):
  """Round 6: in inference mode without a quantiser (the distillation teacher's forward_eval) bn2 / bn3 of a bottleneck block are
  applied in the epilogue of conv1 / conv2 (pf_conv1x1_fwd_affine / pf_conv2d_fwd_affine; graph.Conv2D `out_bn`): the same logits as
  with the stand-alone pass, no bn_apply launch for those layers, conv3 without a prologue -- and nothing changes in training mode."""
  from pocketflow_amd import graph as G
  out = {}
  for fold in (False, True):
    fake = FakeHip()
    monkeypatch.setattr(G, 'hip', fake)
    monkeypatch.setattr(G, 'fusable_tensor', lambda t: True)
    monkeypatch.setattr(G, 'FOLD_EVAL_BN', fold)
    g, net = _build(True, fake, None, filters)
    g.training = False
    g.begin_step = lambda: None
    torch.manual_seed(0)
    x = torch.randn(4, 3, 12, 12).contiguous(memory_format=torch.channels_last)
    with torch.no_grad(), g.as_default():
      logits = net(x, False)
    out[fold] = (logits.clone(), dict(fake.calls))
  a, b = out[False], out[True]
  assert float((a[0] - b[0]).abs().max()) <= 1e-5 * float(a[0].abs().max())
  n_aff = b[1].get('conv_out_affine', 0)
  # conv1 of the four blocks always folds bn2; conv2 folds bn3 where the 3x3 runs on the implicit-GEMM entry point (C % 64 == 0)
  assert n_aff == (8 if filters == 64 and G.OWN_CONV2D else 4) and a[1].get('conv_out_affine', 0) == 0, (a[1], b[1])
  assert b[1].get('bn_apply', 0) == a[1].get('bn_apply', 0) - 4, (a[1], b[1])     # bn2's stand-alone passes are gone (bn3 was lazy anyway)
  # training mode ignores the hint
  fake = FakeHip()
  monkeypatch.setattr(G, 'hip', fake)
  g, net = _build(True, fake, None, filters)
  g.begin_step = lambda: None
  with g.as_default():
    net(x, True).sum().backward()
  assert fake.calls.get('conv_out_affine', 0) == 0


@pytest.mark.parametrize('fuse,filters,size', [(False, 8, 12), (True, 8, 12), (True, 64, 32)])
def test_a_step_leaves_no_reference_cycles(monkeypatch, fuse, filters, size):
  """Every tensor a training step allocates must die by REFERENCE COUNTING when the step ends.  A cycle through an autograd
  node (ctx -> LazyAct -> output alias -> grad_fn -> ctx) is only freed when Python's cyclic collector happens to run: on
  the GPU one step's activations (tens of GB at batch 256) then stay allocated for a random number of further steps, the
  caching allocator grows to the memory ceiling and every allocation turns into a malloc retry (measured: the host-bound
  bench processes of round 2).  Here: collector disabled, several steps, and the set of live tensors must not grow."""
  import gc
  from pocketflow_amd import graph as G
  fake = FakeHip()
  monkeypatch.setattr(G, 'hip', fake)
  monkeypatch.setattr(G, 'fusable_tensor', lambda t: True)
  g, net = _build(fuse, fake, 6, filters)                 # filters = 64: + the implicit-GEMM 3x3 path with its BN box
  g.begin_step = lambda: None
  fake.minmax_slots_init(g.act_slots)
  x = torch.randn(4, 3, size, size).contiguous(memory_format=torch.channels_last)
  wts = torch.randn(4, 7)

  def step():
    with g.as_default():
      logits = net(x, True)
    (logits * wts).sum().backward()
    g.store.zero_grad()

  def live_tensor_bytes():
    return sum(o.numel() * o.element_size() for o in gc.get_objects() if isinstance(o, torch.Tensor))
  step()
  gc.collect()
  gc.disable()
  try:
    step()
    base = live_tensor_bytes()
    for _ in range(3):
      step()
    grown = live_tensor_bytes() - base
  finally:
    gc.enable()
  assert grown <= 0, 'tensors of finished steps are still alive without the cyclic collector: +%d bytes' % grown
  if filters == 64:
    assert fake.calls.get('conv2d_fwd', 0) > 0 and fake.calls.get('conv2d_bwd_data_bnstats', 0) > 0


def test_two_consumer_join_in_backward_data(monkeypatch):
  """bn1 of a projection block has two fused consumers (shortcut convolution + conv1).  With the join, the consumer whose
  backward runs first parks its input gradient and the second one adds it as the residual operand of its backward-data
  launch (no autograd add): same gradients as without the join, one plain launch per projection block replaced by one
  with a residual operand, and the order is the one the block arranges (the dense conv1 joins the strided shortcut)."""
  from pocketflow_amd import graph as G
  results = {}
  for join in (False, True):
    fake = FakeHip()
    monkeypatch.setattr(G, 'hip', fake)
    monkeypatch.setattr(G, 'fusable_tensor', lambda t: True)
    monkeypatch.setattr(G, 'JOIN_TWO_CONSUMERS', join)
    seen = []
    orig = fake.conv1x1_fwd

    def spy(X, W, Y, M, N, K, R=None, scale_shift=None, **kw):
      if scale_shift is None and R is not None:
        seen.append((kw.get('geom') is not None, M, N, K))
      return orig(X, W, Y, M, N, K, R=R, scale_shift=scale_shift, **kw)
    fake.conv1x1_fwd = spy
    g, net = _build(True, fake, None, 8)
    torch.manual_seed(0)
    x = torch.randn(4, 3, 12, 12).contiguous(memory_format=torch.channels_last)
    g.begin_step = lambda: None
    fake.minmax_slots_init(g.act_slots)
    with g.as_default():
      logits = net(x, True)
    n_fwd_res = len(seen)                                  # plain launches with a residual in the forward pass: none
    (logits * torch.randn(4, 7)).sum().backward()
    results[join] = dict(w_grad=g.store.w_grad.clone(), o_grad=g.store.o_grad.clone(), joins=seen[n_fwd_res:])
  assert results[False]['joins'] == [] and len(results[True]['joins']) == 2          # two projection blocks in this net
  assert all(not strided for strided, _, _, _ in results[True]['joins'])            # the DENSE consumer does the join
  for k in ('w_grad', 'o_grad'):
    a, b = results[False][k], results[True][k]
    assert float((a - b).abs().max() / (a.abs().max() + 1e-12)) <= 2e-6, k


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


def test_roofline_numerator_of_the_conv1x1_region(monkeypatch, tmp_path):
  """bench.py's `roofline.achieved` = algorithmic bytes / measured time of the `conv1x1_fwd` region.  The bytes the
  product accounts per launch, (M*K + M*N [+ M*N residual]) * 2 B, are checked here against an
  independent walk over the ResNet-v2-50 bottleneck shapes: one quantised training step on CPU."""
  import contextlib
  import pocketflow_amd.graph as G
  import pocketflow_amd.plan as P
  import pocketflow_amd.losses as L
  import pocketflow_amd.optim as Opt
  import pocketflow_amd.learners.abstract_learner as AL
  import pocketflow_amd.learners.learner_utils  # noqa: F401
  from fake_hip import FakeHipFull
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd.nets.resnet_at_ilsvrc12 import ModelHelper
  from pocketflow_amd.learners.uniform_quantization.learner import UniformQuantLearner
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  fake = FakeHipFull()
  for mod in (G, P, L, Opt):
    monkeypatch.setattr(mod, 'hip', fake)
  monkeypatch.setattr(AL, 'require_gpu', lambda: torch.device('cpu'))
  monkeypatch.setattr(G, 'fusable_tensor', lambda t: True)        # the fused path is for bf16 / fp32 CUDA tensors
  seen = []

  @contextlib.contextmanager
  def region(name, work=0.0):
    seen.append((name, work))
    yield
  monkeypatch.setattr(G, 'region', region)
  B, S = 2, 64
  FLAGS.save_path = str(tmp_path / 'models' / 'model.ckpt')
  FLAGS.uql_save_quant_model_path = str(tmp_path / 'uql' / 'm.ckpt')
  FLAGS.resnet_size, FLAGS.nb_classes, FLAGS.image_size, FLAGS.batch_size, FLAGS.batch_size_eval = 50, 1001, S, B, B
  FLAGS.compute_dtype, FLAGS.synthetic_pool, FLAGS.uql_weight_bits, FLAGS.uql_activation_bits = 'float32', 1, 8, 8
  mh = ModelHelper()
  create_synthetic_checkpoint(mh)
  lrn = UniformQuantLearner(None, mh)
  lrn.train_step()
  got = sorted(w for n, w in seen if n == 'conv1x1_fwd')
  # independent enumeration: stem 7x7/2 + max-pool/2 -> S/4; stages of (3, 4, 6, 3) bottlenecks, 64..512 filters
  want, h, cin = [], S // 4, 64
  for nblk, f, s in ((3, 64, 1), (4, 128, 2), (6, 256, 2), (3, 512, 2)):
    for b in range(nblk):
      stride = s if b == 0 else 1
      ho = h // stride
      if b == 0:
        want.append((B * ho * ho * cin + B * ho * ho * 4 * f) * 2)            # projection shortcut (strided read)
      want.append((B * h * h * cin + B * h * h * f) * 2)                      # conv1
      want.append((B * ho * ho * f + 2 * B * ho * ho * 4 * f) * 2)            # conv3 + residual read
      h, cin = ho, 4 * f
  assert len(got) == len(want) == 4 + 2 * 16 and got == sorted(float(w) for w in want)
  # the backward regions account the same tensors once per gradient GEMM
  assert len([1 for n, _ in seen if n == 'conv1x1_wrw']) == 36 and len([1 for n, _ in seen if n == 'conv1x1_bwd_data']) == 36


def test_backward_filter_queue_is_the_one_queue_without_a_hip_device():
  """graph.WrwSide (round 6) forks the backward-filter launches of a pass onto a second HIP stream.  Off the device -- the CPU emulation
  of these tests, a store that was never finalised -- arming is a no-op, `_wrw_queue` hands out the graph's own split workspace and no
  side object is ever made; PF_WRW_SIDE=0 (G.WRW_SIDE False) gives the same on a device."""
  import types
  import pocketflow_amd.graph as G
  made = []
  store = types.SimpleNamespace(device=torch.device('cpu'))
  graph = types.SimpleNamespace(store=store, scratch=lambda n: made.append(n) or torch.empty(n))
  with G.wrw_side_armed(store):
    with G._wrw_queue(graph, True, torch.zeros(1)) as scratch:
      assert scratch(7).numel() == 7
  assert made == [7] and not hasattr(store, 'wrw_side')
  with G.wrw_side_armed(None):                       # an optimiser without a store (never happens in the learners; must not raise)
    pass
  old = G.WRW_SIDE
  try:
    G.WRW_SIDE = False
    fake_cuda = types.SimpleNamespace(device=torch.device('cuda', 0))
    with G.wrw_side_armed(fake_cuda):                # switched off: no stream is created even for a device store
      pass
    assert not hasattr(fake_cuda, 'wrw_side')
  finally:
    G.WRW_SIDE = old
