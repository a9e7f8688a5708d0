"""HIP kernels (through the C ABI) vs the NumPy oracle, same seeded inputs.

Integer / index / mask / fake-quant work: bit-exact.  Transcendental or reduction-order dependent
floating point (losses, BN statistics, codebook gradients, GEMM): tolerance stated per test.
"""
import numpy as np
import pytest
import torch

from oracle import pf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
  from pocketflow_amd import hip as h
  assert torch.cuda.is_available(), 'gpu tests need a GPU'
  return h


def dev(a, dtype=None):
  t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
  return t if dtype is None else t.to(dtype)


def new_slot(hip, n=1):
  s = torch.empty((n, 2), dtype=torch.int32, device='cuda')
  hip.minmax_slots_init(s)
  return s


# ------------------------------------------------------------------------------------------------
# K1/K2/K4: per-tensor min/max + uniform fake-quant
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n', [1, 7, 8, 1000, 4099, 262147])
@pytest.mark.parametrize('bits', [1, 2, 4, 8, 32])
@pytest.mark.parametrize('act', [None, 'Relu', 'Relu6'])
def test_uq_apply_bit_exact(hip, n, bits, act):
  rng = np.random.RandomState(n * 31 + bits)
  x = (rng.randn(n) * 3).astype(np.float32)
  t = O.ACTIVATIONS[act](x) if act else x
  ref, info = O.uniform_quantize(t, bits, mode='activation')
  xd = dev(x)
  slot = new_slot(hip)
  hip.minmax_tensor(xd, slot[0], act)
  ab = hip.minmax_decode(slot).cpu().numpy()
  assert ab[0, 0] == info['alpha'] and ab[0, 1] == info['beta']
  y = torch.empty_like(xd)
  hip.uq_apply(xd, y, slot[0], bits, act)
  np.testing.assert_array_equal(y.cpu().numpy(), ref)


def test_uq_half_even_ties_and_constant_tensor(hip):
  # x_hat * k hits exactly .5 / 1.5 / 2.5 -> round-half-to-even
  # bits = 2 -> k = 3 ; values chosen so that x_hat*k = j + 0.5
  x = np.array([0.0, 1.0] + [(j + 0.5) / 3.0 for j in range(3)], dtype=np.float32)
  ref, _ = O.uniform_quantize(x, 2, mode='activation')
  xd = dev(x)
  slot = new_slot(hip)
  hip.minmax_tensor(xd, slot[0])
  y = torch.empty_like(xd)
  hip.uq_apply(xd, y, slot[0], 2)
  np.testing.assert_array_equal(y.cpu().numpy(), ref)
  # max == min: alpha = 1e-10, x_hat = 0 -> y = beta
  c = np.full(100, 0.37, dtype=np.float32)
  ref, _ = O.uniform_quantize(c, 8, mode='activation')
  cd = dev(c)
  slot = new_slot(hip)
  hip.minmax_tensor(cd, slot[0])
  y = torch.empty_like(cd)
  hip.uq_apply(cd, y, slot[0], 8)
  np.testing.assert_array_equal(y.cpu().numpy(), ref)


def test_uq_apply_bf16_storage(hip):
  rng = np.random.RandomState(5)
  x32 = (rng.randn(70001) * 2).astype(np.float32)
  xb = dev(x32).to(torch.bfloat16)
  xr = xb.float().cpu().numpy()                       # the values the kernel actually sees
  ref, _ = O.uniform_quantize(np.maximum(xr, 0), 8, mode='activation')
  slot = new_slot(hip)
  hip.minmax_tensor(xb, slot[0], 'Relu')
  y = torch.empty_like(xb)
  hip.uq_apply(xb, y, slot[0], 8, 'Relu')
  expect = torch.from_numpy(ref).to(torch.bfloat16)   # RNE cast of the fp32 result
  assert torch.equal(y.cpu(), expect)


def test_act_grad(hip):
  rng = np.random.RandomState(9)
  for act in ('Relu', 'Relu6'):
    u = (rng.randn(5003) * 4).astype(np.float32)
    u[:3] = [0.0, 6.0, -0.0]
    g = rng.randn(5003).astype(np.float32)
    dx = torch.empty(5003, device='cuda')
    hip.act_grad(dev(g), dev(u), dx, act)
    np.testing.assert_array_equal(dx.cpu().numpy(), O.activation_quantize_grad(g, u, act))


# ------------------------------------------------------------------------------------------------
# segment kernels: all weight tensors at once, per-tensor / channel / split buckets
# ------------------------------------------------------------------------------------------------
WEIGHT_SHAPES = [((3, 3, 5, 7), 'conv'), ((1, 1, 64, 256), 'conv'), ((7, 7, 3, 64), 'conv'),
                 ((3, 3, 128, 128), 'conv'), ((1600, 256), 'dense'), ((3, 3, 32, 1), 'depthwise'),
                 ((1, 1, 16, 8), 'conv'), ((130, 10), 'dense')]


def make_store(shapes, seed=0):
  from pocketflow_amd.graph import VarStore
  st = VarStore('model')
  rng = np.random.RandomState(seed)
  vals = {}
  for i, (shp, kind) in enumerate(shapes):
    v = st.add('w%d/kernel' % i, shp, kind)
    vals[v.name] = (rng.randn(*shp) * (0.1 + i)).astype(np.float32)
  st.finalize('cuda', torch.float32, separate_compute=True)
  st.load_numpy(vals)
  return st, vals


@pytest.mark.parametrize('use_buckets,bucket_type,bucket_size',
                         [(False, 'channel', 256), (True, 'channel', 256), (True, 'split', 256),
                          (True, 'split', 4), (True, 'split', 1000)])
@pytest.mark.parametrize('bits', [2, 8])
def test_seg_uniform_quantize_bit_exact(hip, use_buckets, bucket_type, bucket_size, bits):
  from pocketflow_amd.plan import QuantPlan
  st, vals = make_store(WEIGHT_SHAPES)
  vars_ = st.matmul_vars
  bit_list = [bits] * len(vars_)
  bit_list[1] = 0                                     # one tensor left un-quantised (plain copy)
  plan = QuantPlan(st.weight_descs(vars_), bit_list, use_buckets, bucket_type, bucket_size, st.device)
  plan.uniform_quantize(st.w_master, st.w_compute)
  for v, b in zip(vars_, bit_list):
    got = v.to_ref(st.w_compute[v.offset:v.offset + v.numel].cpu().numpy())
    if b == 0:
      np.testing.assert_array_equal(got, vals[v.name])
      continue
    ref, info = O.uniform_quantize(vals[v.name], b, 'weight', use_buckets, bucket_type, bucket_size)
    np.testing.assert_array_equal(got, ref, err_msg=v.name)
  if use_buckets:
    assert plan.bucket_storage_bits == sum(
        O.uniform_quantize(vals[v.name], b, 'weight', True, bucket_type, bucket_size)[1]['bucket_storage_bits']
        for v, b in zip(vars_, bit_list) if b)


def test_seg_quantize_bf16_output_is_rne_cast(hip):
  from pocketflow_amd.graph import VarStore
  from pocketflow_amd.plan import QuantPlan
  st = VarStore('model')
  v = st.add('c/kernel', (3, 3, 16, 24), 'conv')
  st.finalize('cuda', torch.bfloat16)
  w = np.random.RandomState(1).randn(3, 3, 16, 24).astype(np.float32)
  st.load_numpy({v.name: w})
  plan = QuantPlan(st.weight_descs([v]), [4], False, 'channel', 256, st.device)
  plan.uniform_quantize(st.w_master, st.w_compute)
  ref, _ = O.uniform_quantize(w, 4)
  expect = torch.from_numpy(v.to_storage(ref)).to(torch.bfloat16).reshape(-1)
  assert torch.equal(st.w_compute[:v.numel].cpu(), expect)


@pytest.mark.parametrize('use_buckets,bucket_type,bucket_size',
                         [(False, 'split', 256), (True, 'split', 256), (True, 'channel', 256), (True, 'split', 8)])
@pytest.mark.parametrize('bits', [1, 2, 4])
def test_seg_nonuniform_quantize(hip, use_buckets, bucket_type, bucket_size, bits):
  from pocketflow_amd.plan import QuantPlan
  shapes = WEIGHT_SHAPES[:5] + WEIGHT_SHAPES[6:]
  st, vals = make_store(shapes, seed=3)
  vars_ = st.matmul_vars
  plan = QuantPlan(st.weight_descs(vars_), [bits] * len(vars_), use_buckets, bucket_type, bucket_size,
                   st.device, nuq=True)
  k = 2 ** bits
  cbs = np.zeros(plan.n_codebook, dtype=np.float32)
  infos = []
  for s, v in enumerate(vars_):
    ref, info = O.nuq_quantize(vals[v.name], bits, None, use_buckets, bucket_type, bucket_size)
    infos.append((ref, info))
    cb = info['codebook'].reshape(k, -1)
    # channel bucket on a dense / conv tensor: n_bucket = cout
    cbs[plan.cb_offsets[s]:plan.cb_offsets[s] + cb.size] = cb.reshape(-1)
  cbd = dev(cbs)
  idx = torch.zeros(st.w_size, dtype=torch.uint8, device='cuda')
  plan.nonuniform_quantize(st.w_master, st.w_compute, idx, cbd)
  g_flat = torch.zeros(st.w_size, device='cuda')
  gvals = {}
  rng = np.random.RandomState(11)
  for v in vars_:
    g = rng.randn(*v.ref_shape).astype(np.float32)
    gvals[v.name] = g
    g_flat[v.offset:v.offset + v.numel] = dev(v.to_storage(g)).reshape(-1)
  dcb = torch.zeros(plan.n_codebook, device='cuda')
  plan.codebook_grad(g_flat, idx, dcb)
  # bit-reproducible: 64-bit integer (fixed-point) atomics, VERDICT r1 next-step 8
  for _ in range(2):
    dcb2 = torch.zeros(plan.n_codebook, device='cuda')
    plan.codebook_grad(g_flat, idx, dcb2)
    assert torch.equal(dcb, dcb2)
  dcb = dcb.cpu().numpy()
  for s, v in enumerate(vars_):
    ref, info = infos[s]
    got = v.to_ref(st.w_compute[v.offset:v.offset + v.numel].cpu().numpy())
    np.testing.assert_array_equal(got, ref, err_msg=v.name)
    _, dc_ref = O.nuq_backward(gvals[v.name], info, use_buckets, bucket_type, bucket_size)
    got_dc = dcb[plan.cb_offsets[s]:plan.cb_offsets[s] + dc_ref.size].reshape(dc_ref.reshape(k, -1).shape)
    # exact fixed-point sums (one 2^-36 rounding per term) vs the oracle's float32 sums
    np.testing.assert_allclose(got_dc, dc_ref.reshape(k, -1), rtol=1e-4,
                               atol=1e-4 * np.abs(dc_ref).max() + 1e-6)


def test_seg_normalize_matches_scale(hip):
  from pocketflow_amd.plan import QuantPlan
  st, vals = make_store(WEIGHT_SHAPES[:3])
  vars_ = st.matmul_vars
  plan = QuantPlan(st.weight_descs(vars_), [4] * 3, False, 'split', 256, st.device, nuq=True)
  plan.calibrate(st.w_master)
  for s, v in enumerate(vars_):
    xn = torch.empty(v.numel, device='cuda')
    hip.seg_normalize(st.w_master, xn, plan.segs, s, plan.slots)
    ref, _, _ = O.scale(vals[v.name], None)
    np.testing.assert_array_equal(v.to_ref(xn.cpu().numpy()), ref)


# ------------------------------------------------------------------------------------------------
# weight sparsification / channel pruning
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n', [1, 2, 10, 11, 1000, 271153])
def test_kth_largest_matches_percentile(hip, n):
  rng = np.random.RandomState(n)
  a = np.abs(rng.randn(n)).astype(np.float32)
  if n > 10:
    a[::7] = a[3]                                      # ties
  ad = dev(a)
  out = torch.empty(1, device='cuda')
  ws = torch.empty(1024, dtype=torch.int32, device='cuda')
  for q in [0.0, 12.5, 50.0, 75.0, 99.9, 100.0]:
    ref = O.percentile_nearest(a, q)
    d = n
    idx = int(np.clip(np.rint((d - 1) * (1.0 - q / 100.0)), 0, d - 1))
    hip.kth_largest_nonneg(ad, idx, out, ws)
    assert out.item() == ref, (n, q)


def test_ws_mask_refresh_two_rounds(hip):
  rng = np.random.RandomState(2)
  n = 50000
  var = rng.randn(n).astype(np.float32)
  bkup = var.copy()
  mask = np.ones(n, dtype=np.float32)
  vd, bd, md = dev(var), dev(bkup), dev(mask)
  absb = torch.empty(n, device='cuda')
  thr = torch.empty(1, device='cuda')
  ws = torch.empty(1024, dtype=torch.int32, device='cuda')
  for rnd, r in enumerate([0.3, 0.5]):
    var, bkup, mask, t = O.ws_mask_refresh(var, bkup, mask, np.float32(r))
    hip.ws_bkup_merge_abs(vd, bd, md, absb)
    q = np.float64(np.float32(np.float32(r) * np.float32(100.0)))
    idx = int(np.clip(np.rint((n - 1) * (1.0 - q / 100.0)), 0, n - 1))
    hip.kth_largest_nonneg(absb, idx, thr, ws)
    hip.ws_mask_apply(vd, bd, md, thr)
    assert thr.item() == t
    np.testing.assert_array_equal(md.cpu().numpy(), mask)
    np.testing.assert_array_equal(vd.cpu().numpy(), var)
    np.testing.assert_array_equal(bd.cpu().numpy(), bkup)
    # live weights drift between refreshes; pruned ones stay zero in var but keep their backup
    upd = rng.randn(n).astype(np.float32) * 0.1 * mask
    var = (var + upd).astype(np.float32)
    vd = dev(var)
  cnt = torch.zeros(1, dtype=torch.int64, device='cuda')
  hip.count_nonzero(vd, cnt)
  assert cnt.item() == np.count_nonzero(var)


def test_cp_masks(hip):
  rng = np.random.RandomState(4)
  O_, RS, I = 24, 9, 40
  keep_in = rng.rand(I) > 0.4
  keep_out = rng.rand(O_) > 0.3
  ref = O.cp_grad_mask((3, 3, I, O_), keep_in, keep_out)            # HWIO
  m = torch.empty(O_ * RS * I, device='cuda')
  ki, ko = dev(keep_in.astype(np.uint8)), dev(keep_out.astype(np.uint8))
  hip.cp_build_mask(m, ki, ko, O_, RS, I)
  np.testing.assert_array_equal(m.view(O_, 3, 3, I).permute(1, 2, 3, 0).cpu().numpy(), ref)
  g = rng.randn(O_, 3, 3, I).astype(np.float32)
  gd = dev(g).reshape(-1).clone()
  hip.cp_mask_grad(gd, ki, ko, O_, RS, I)
  np.testing.assert_array_equal(gd.view(O_, 3, 3, I).cpu().numpy(), g * np.transpose(ref, (3, 0, 1, 2)))


# ------------------------------------------------------------------------------------------------
# optimisers
# ------------------------------------------------------------------------------------------------
def test_adam_flat_matches_oracle_trace(hip):
  rng = np.random.RandomState(6)
  n, n_decay, wd = 4099, 3000, 1e-4
  p = rng.randn(n).astype(np.float32)
  m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
  mask = (rng.rand(n) > 0.3).astype(np.float32)
  pad = (-n) % 4
  pd, md_, vd = dev(np.pad(p, (0, pad))), dev(np.pad(m, (0, pad))), dev(np.pad(v, (0, pad)))
  maskd = dev(np.pad(mask, (0, pad)))
  b1p, b2p = np.float32(0.9), np.float32(0.999)
  for t in range(1, 4):
    g = rng.randn(n).astype(np.float32)
    ge = g.copy()
    ge[:n_decay] = ge[:n_decay] + np.float32(wd) * p[:n_decay]
    ge = ge * mask
    p, m, v = O.adam_step(p, ge, m, v, t, 1e-3)
    hip.adam_flat(pd[:n], dev(np.pad(g, (0, pad)))[:n], md_[:n], vd[:n], maskd[:n], n_decay, wd, 1.0, 1e-3, 0.9,
                  0.999, 1e-8, float(b1p), float(b2p))
    b1p = np.float32(b1p * np.float32(0.9)); b2p = np.float32(b2p * np.float32(0.999))
    np.testing.assert_allclose(pd[:n].cpu().numpy(), p, rtol=2e-7, atol=1e-9)
    np.testing.assert_allclose(md_[:n].cpu().numpy(), m, rtol=2e-7, atol=1e-12)
    np.testing.assert_allclose(vd[:n].cpu().numpy(), v, rtol=2e-7, atol=1e-12)


def test_momentum_flat_bit_exact(hip):
  rng = np.random.RandomState(7)
  n = 1024
  p = rng.randn(n).astype(np.float32); acc = np.zeros(n, np.float32)
  pd, ad = dev(p), dev(acc)
  for t in range(3):
    g = rng.randn(n).astype(np.float32)
    p, acc = O.momentum_step(p, g, acc, 0.1, 0.9)
    hip.momentum_flat(pd, dev(g), ad, None, 0, 0.0, 1.0, 0.1, 0.9)
    np.testing.assert_array_equal(pd.cpu().numpy(), p)
    np.testing.assert_array_equal(ad.cpu().numpy(), acc)


def test_adam_bf16_grads_and_allreduce_scale(hip):
  rng = np.random.RandomState(8)
  n = 2048
  p = rng.randn(n).astype(np.float32)
  g = rng.randn(n).astype(np.float32)
  gb = dev(g).to(torch.bfloat16)
  gr = gb.float().cpu().numpy()
  p_ref, m_ref, v_ref = O.adam_step(p, np.float32(0.5) * gr, np.zeros(n, np.float32), np.zeros(n, np.float32), 1, 1e-2)
  pd, md_, vd = dev(p), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
  hip.adam_flat(pd, gb, md_, vd, None, 0, 0.0, 0.5, 1e-2, 0.9, 0.999, 1e-8, 0.9, 0.999)
  np.testing.assert_allclose(pd.cpu().numpy(), p_ref, rtol=2e-7, atol=1e-9)


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,C', [(4, 10), (32, 1001), (3, 2500)])
def test_ce_distill_fused(hip, B, C):
  rng = np.random.RandomState(B + C)
  zs = (rng.randn(B, C) * 3).astype(np.float32)
  zt = (rng.randn(B, C) * 3).astype(np.float32)
  lab = np.zeros((B, C), np.float32); lab[np.arange(B), rng.randint(0, C, B)] = 1
  ce, dce = O.softmax_cross_entropy(lab, zs)
  dl, ddl = O.distill_loss(zs, zt, 4.0, 4.0)
  losses = torch.empty(2, device='cuda'); dz = torch.empty((B, C), device='cuda'); ws = torch.empty(2 * B, device='cuda')
  hip.ce_distill_fwd_bwd(dev(zs), dev(lab), dev(zt), 4.0, 4.0, losses, dz, ws)
  got = losses.cpu().numpy()
  # tolerance: 1e-5 relative (expf/logf differ by a few ulp from NumPy's)
  np.testing.assert_allclose(got, [ce, dl], rtol=1e-5)
  np.testing.assert_allclose(dz.cpu().numpy(), dce + ddl, rtol=1e-4, atol=1e-7)
  # no teacher
  hip.ce_distill_fwd_bwd(dev(zs), dev(lab), None, 1.0, 0.0, losses, dz, ws)
  np.testing.assert_allclose(losses.cpu().numpy(), [ce, 0.0], rtol=1e-5)
  np.testing.assert_allclose(dz.cpu().numpy(), dce, rtol=1e-4, atol=1e-7)
  # closed form of the distillation gradient: w/(B*T) * (softmax(zs/T) - softmax(zt/T))
  closed = 4.0 / (B * 4.0) * (O.softmax(zs / 4.0) - O.softmax(zt / 4.0))
  np.testing.assert_allclose(ddl, closed, rtol=1e-4, atol=1e-7)


# ------------------------------------------------------------------------------------------------
# fused BN + act + fake-quant
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('rows,C', [(64, 16), (1000, 64), (777, 24), (4096, 256)])
@pytest.mark.parametrize('act,bits', [('Relu', 8), ('Relu6', 4), ('Relu', None)])
def test_bn_act_quant_forward_backward(hip, rows, C, act, bits):
  from pocketflow_amd.graph import BatchNormAct, Graph
  rng = np.random.RandomState(rows + C)
  x = (rng.randn(rows, C) * 2 + rng.randn(C) * 3).astype(np.float32)
  gamma = (rng.rand(C) + 0.5).astype(np.float32) * np.where(rng.rand(C) > 0.8, -1, 1).astype(np.float32)
  beta = rng.randn(C).astype(np.float32)
  g = Graph('model', 'cuda', torch.float32)
  layer = BatchNormAct(g, 'bn', C, act, 0.997, 1e-5)
  g.finalize()
  g.store.load_numpy({'model/bn/gamma': gamma, 'model/bn/beta': beta}, strict=False)
  layer.op.bits = bits
  g.begin_step()
  g.training = True
  xd = dev(x).view(rows, C, 1, 1).requires_grad_(True)
  q = layer(xd)
  # oracle
  y, mm, mv, (mean, inv_std) = O.batch_norm_train(x, gamma, beta, np.zeros(C), np.ones(C), 0.997, 1e-5)
  t = O.ACTIVATIONS[act](y)
  # un-quantised output of the same kernels (second graph, no fake-quant behind the activation)
  g2 = Graph('model', 'cuda', torch.float32)
  l2 = BatchNormAct(g2, 'bn', C, act, 0.997, 1e-5); g2.finalize()
  g2.store.load_numpy({'model/bn/gamma': gamma, 'model/bn/beta': beta}, strict=False)
  g2.begin_step(); g2.training = True
  y_gpu = l2(dev(x).view(rows, C, 1, 1)).detach().view(rows, C)
  # BN statistics are reduction-order dependent: 2e-5 absolute on O(1) values
  np.testing.assert_allclose(y_gpu.cpu().numpy(), t, rtol=1e-4, atol=2e-5)
  if bits is None:
    assert torch.equal(q.detach().view(rows, C), y_gpu)
  else:
    # the activation range is the exact whole-tensor min/max of what the kernel itself produced
    ab = g.act_alpha_beta().cpu().numpy()[0]
    mn, mx = y_gpu.min().item(), y_gpu.max().item()
    assert ab[1] == np.float32(mn) and ab[0] == np.float32(np.float32(mx) - np.float32(mn)) + np.float32(1e-10)
    ref_q, _ = O.uniform_quantize(y_gpu.cpu().numpy(), bits, mode='activation')
    np.testing.assert_array_equal(q.detach().view(rows, C).cpu().numpy(), ref_q)
  np.testing.assert_allclose(layer.moving_mean.tensor.cpu().numpy(), mm, rtol=1e-4, atol=1e-6)
  np.testing.assert_allclose(layer.moving_var.tensor.cpu().numpy(), mv, rtol=1e-4, atol=1e-6)
  # backward
  dq = rng.randn(rows, C).astype(np.float32)
  q.backward(dev(dq).view(rows, C, 1, 1))
  # ReLU mask from the GPU's own pre-activation sign (elements with |y| ~ 1e-7 may differ from fp64)
  yg = y_gpu.cpu().numpy()
  dy = dq * ((yg > 0) if act == 'Relu' else ((yg > 0) & (yg < 6))).astype(np.float32)
  dx_ref, dg_ref, db_ref = O.batch_norm_train_bwd(dy, x, gamma, mean, inv_std)
  scale = np.abs(dx_ref).max()
  np.testing.assert_allclose(xd.grad.view(rows, C).cpu().numpy(), dx_ref, rtol=1e-3, atol=1e-4 * scale)
  np.testing.assert_allclose(layer.gamma.tensor.grad.cpu().numpy(), dg_ref, rtol=1e-3, atol=1e-3 * np.abs(dg_ref).max())
  np.testing.assert_allclose(layer.beta.tensor.grad.cpu().numpy(), db_ref, rtol=1e-3, atol=1e-3 * np.abs(db_ref).max())


def test_bn_eval_mode(hip):
  from pocketflow_amd.graph import BatchNormAct, Graph
  rng = np.random.RandomState(0)
  rows, C = 500, 32
  x = rng.randn(rows, C).astype(np.float32)
  vals = {'model/bn/gamma': rng.rand(C).astype(np.float32) + .5, 'model/bn/beta': rng.randn(C).astype(np.float32),
          'model/bn/moving_mean': rng.randn(C).astype(np.float32), 'model/bn/moving_variance': rng.rand(C).astype(np.float32) + .5}
  g = Graph('model', 'cuda', torch.float32)
  layer = BatchNormAct(g, 'bn', C, 'Relu', 0.997, 1e-5); g.finalize(); g.store.load_numpy(vals)
  g.training = False
  with torch.no_grad():
    y = layer(dev(x).view(rows, C, 1, 1)).view(rows, C).cpu().numpy()
  ref = np.maximum(O.batch_norm_eval(x, vals['model/bn/gamma'], vals['model/bn/beta'], vals['model/bn/moving_mean'],
                                     vals['model/bn/moving_variance'], 1e-5), 0)
  np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-5)
  np.testing.assert_array_equal(layer.moving_mean.tensor.cpu().numpy(), vals['model/bn/moving_mean'])


# ------------------------------------------------------------------------------------------------
# full-size, size-independent properties (ResNet-50 @ batch 256 shapes; no oracle at this size)
# ------------------------------------------------------------------------------------------------
def test_fullsize_activation_quant_properties(hip):
  n = 64 * 56 * 56 * 256                               # a quarter of the largest R50 activation
  x = torch.randn(n, device='cuda', dtype=torch.bfloat16) * 3
  slot = new_slot(hip)
  hip.minmax_tensor(x, slot[0], 'Relu')
  ab = hip.minmax_decode(slot).cpu().numpy()[0]
  r = torch.relu(x.float())
  assert ab[1] == r.min().item() and np.float32(ab[0]) == np.float32(np.float32(r.max().item()) - np.float32(r.min().item())) + np.float32(1e-10)
  y = torch.empty_like(x)
  hip.uq_apply(x, y, slot[0], 8, 'Relu')
  assert torch.unique(y).numel() <= 256                 # at most 2^bits levels
  assert y.min().item() >= 0 and y.float().max().item() <= r.max().item() * (1 + 1e-2)
  # monotone: sorting commutes with quantisation
  xs, _ = torch.sort(x[:1 << 20].float())
  ys = torch.empty(1 << 20, device='cuda')
  hip.uq_apply(xs.contiguous(), ys, slot[0], 8, 'Relu')
  assert bool((ys[1:] >= ys[:-1]).all())
  # idempotent in fp32: re-quantising grid values with the same range returns them unchanged
  xf = r[:1 << 20].contiguous()
  y1 = torch.empty_like(xf); y2 = torch.empty_like(xf)
  hip.uq_apply(xf, y1, slot[0], 8)
  hip.uq_apply(y1, y2, slot[0], 8)
  assert torch.equal(y1, y2)


# ---------------------------------------------------------------------------------------------------------------------
# K13: stem max-pooling (tf.layers.max_pooling2d, padding SAME) and its gradient
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('special', ['1', '0'])            # PF_POOL3S2: the 3x3 / stride-2 kernels of round 3 vs the generic pair
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,C,H,W,k,stride', [(3, 64, 112, 112, 3, 2), (2, 16, 15, 9, 3, 2), (5, 8, 16, 9, 3, 2), (1, 8, 1, 2, 3, 2),
                                              (2, 8, 8, 8, 2, 2), (1, 24, 7, 7, 3, 1)])
def test_maxpool_same_forward_and_gradient_match_torch(monkeypatch, special, dtype, B, C, H, W, k, stride):
  """Clipped windows == -inf padding (extra pixel at the END, TF 'SAME'); the gradient goes to the FIRST maximum of a
  window.  Inputs are drawn from a small integer set so that ties inside a window are common."""
  import torch.nn.functional as F
  from pocketflow_amd import graph as G
  if special == '0' and not (k == 3 and stride == 2):
    pytest.skip('generic kernels either way')
  monkeypatch.setenv('PF_POOL3S2', special)
  g = torch.Generator(device='cuda').manual_seed(B * C + H)
  x = torch.randint(-3, 4, (B, C, H, W), device='cuda', generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
  x.requires_grad_(True)
  y = G.max_pool_same(x, k, stride)
  ph, pw = G._same_pads(H, k, stride), G._same_pads(W, k, stride)
  xr = x.detach().float().requires_grad_(True)
  ref = F.max_pool2d(F.pad(xr, (pw[0], pw[1], ph[0], ph[1]), value=float('-inf')), k, stride)
  assert y.shape == ref.shape and torch.equal(y.float(), ref)
  dy = torch.randint(-2, 3, ref.shape, device='cuda', generator=g).float()
  y.backward(dy.to(dtype).contiguous(memory_format=torch.channels_last))
  ref.backward(dy)
  assert torch.equal(x.grad.float(), xr.grad)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_seg_transpose_gives_the_backward_data_layout_of_every_kernel(dtype):
  """W'[c][R-1-r][S-1-s][n] = W[n][r][s][c] for all convolution kernels of a store in one launch (pf_seg_transpose)."""
  from pocketflow_amd.graph import VarStore
  st = VarStore('model')
  shapes = [(1, 1, 64, 256), (3, 3, 64, 64), (1, 1, 72, 40), (3, 3, 130, 70), (7, 7, 3, 64), (5, 3, 16, 8)]
  vs = [st.add('c%d/kernel' % i, s, 'conv') for i, s in enumerate(shapes)]
  st.add('fc/kernel', (100, 10), 'dense')
  st.finalize('cuda', dtype)
  rng = np.random.RandomState(0)
  st.load_numpy({v.name: rng.randn(*v.ref_shape).astype(np.float32) for v in st.vars if v.group == 'W'})
  for v in vs:
    got = st.transposed(v)
    w_krsc = v.tensor.detach().permute(0, 2, 3, 1)                       # [N][R][S][C]
    ref = w_krsc.flip(1, 2).permute(3, 1, 2, 0).contiguous()              # [C][R][S][N]
    assert got.shape == ref.shape and torch.equal(got, ref), v.name
  # stale after the compute copy changes, fresh again on the next request
  st.w_master.mul_(2.0)
  st.sync_compute()
  assert not st.w_t_fresh
  assert torch.equal(st.transposed(vs[1]), vs[1].tensor.detach().permute(0, 2, 3, 1).flip(1, 2).permute(3, 1, 2, 0).contiguous())


def test_launch_stream_is_torchs_current_stream(hip):
  """hip._stream() (the raw getter) must follow torch's stream context exactly: the default stream, a side stream, and a stream
  being captured into a graph (tools/gpu/_timing.py launches the kernels there)."""
  assert hip._stream().value in (torch.cuda.current_stream().cuda_stream, None if torch.cuda.current_stream().cuda_stream == 0 else -1) \
      or (hip._stream().value or 0) == torch.cuda.current_stream().cuda_stream
  s = torch.cuda.Stream()
  with torch.cuda.stream(s):
    assert (hip._stream().value or 0) == s.cuda_stream
  assert (hip._stream().value or 0) == torch.cuda.current_stream().cuda_stream
  # a kernel launched on the side stream is ordered with that stream's work
  x = torch.zeros(1 << 20, device='cuda')
  slot = torch.empty(2, dtype=torch.int32, device='cuda')
  s.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(s):
    x.add_(3.0)
    hip.minmax_slots_init(slot)
    hip.minmax_tensor(x, slot)
  s.synchronize()
  ab = hip.minmax_decode(slot).cpu().numpy()
  assert abs(float(ab[0, 1]) - 3.0) < 1e-6                       # beta = min = 3: the reduction saw the add


@pytest.mark.parametrize('O_,k,I,gdt', [(64, 3, 64, torch.float32), (256, 1, 512, torch.bfloat16), (10, 5, 24, torch.float32), (512, 3, 256, torch.bfloat16)])
def test_cpg_proximal_step_matches_oracle(O_, k, I, gdt):
  """pf_prox_norms -> pf_kth_largest_nonneg -> pf_prox_apply (the 'chn-pruned-gpu' learner's proximal-gradient step, pf_prox.hip)
  against oracle/pf_oracle.py cpg_proximal_step on the same float32 kernel and gradient: channel norms to float32 summation order,
  the threshold IS one of the device's norms (nearest rank), the shrunk kernel to 2e-6, pruned channels exactly zero; a second
  run gives the same bits."""
  from oracle import pf_oracle as O
  from pocketflow_amd import hip
  from pocketflow_amd.learners.channel_pruning_gpu.learner import proximal_step
  rng = np.random.RandomState(O_ + I)
  w_hwio = (rng.randn(k, k, I, O_) * (0.2 + rng.rand(1, 1, I, 1))).astype(np.float32)
  g_hwio = rng.randn(k, k, I, O_).astype(np.float32)
  if gdt == torch.bfloat16:
    g_hwio = torch.from_numpy(g_hwio).bfloat16().float().numpy()
  w_hwio[:, :, 3, :] = 0
  g_hwio[:, :, 3, :] = 0
  rows = O_ * k * k
  to_krsc = lambda a: np.ascontiguousarray(a.transpose(3, 0, 1, 2)).reshape(-1)
  for lr, perctl in ((1e-3, 0.0), (2e-2, 30.0), (1e-2, 50.0)):
    want, norm, thr = O.cpg_proximal_step(w_hwio, g_hwio, lr, perctl)
    outs = []
    for rep in range(2):
      w = torch.from_numpy(to_krsc(w_hwio)).cuda()
      g = torch.from_numpy(to_krsc(g_hwio)).cuda().to(gdt)
      ws = (torch.full((hip.prox_groups(rows, I) * I,), float('nan'), device='cuda'), torch.full((max(I, 4096),), float('nan'), device='cuda'),
            torch.empty(1, device='cuda'), torch.empty(4096, dtype=torch.int32, device='cuda'))
      proximal_step(w, g, lr, perctl, rows, I, ws)
      outs.append((w.cpu().numpy(), ws[1][:I].cpu().numpy(), float(ws[2][0])))
    got, n, t = outs[0]
    assert np.array_equal(got, outs[1][0]) and t == outs[1][2]
    np.testing.assert_allclose(n, norm, rtol=3e-6)
    assert t in set(n.tolist()) and abs(t - float(thr)) <= 3e-6 * max(1.0, float(thr))
    got_hwio = got.reshape(O_, k, k, I).transpose(1, 2, 3, 0)
    np.testing.assert_allclose(got_hwio, want, rtol=5e-5, atol=3e-6 * float(np.abs(want).max()))
    assert np.array_equal(np.all(got_hwio == 0, axis=(0, 1, 3)), np.all(want == 0, axis=(0, 1, 3)))
