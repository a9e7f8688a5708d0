"""The CPU oracle (oracle/pf_oracle.py) against fixtures produced by EXECUTING the reference's own
hot-path functions (tests/golden/make_reference_golden.py: the functions are lifted from
/root/reference with `ast` and run over oracle/tf_stub.py, a NumPy-eager stand-in for the TF ops).
Everything here is bit-exact: same float32 op order or it fails.  Runs without a GPU."""
import json
import os

import numpy as np
import pytest

from oracle import pf_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def arrays():
  with np.load(os.path.join(GOLD, 'reference_arrays.npz')) as z:
    return {k: z[k] for k in z.files}


@pytest.fixture(scope='module')
def host():
  with open(os.path.join(GOLD, 'reference_host.json')) as f:
    return json.load(f)


UQ_MODES = {'tensor': (False, 'split', 0), 'channel': (True, 'channel', 0), 'split256': (True, 'split', 256),
            'split4': (True, 'split', 4)}


def _cases(arrays, prefix):
  return sorted({k.split('/')[1] for k in arrays if k.startswith(prefix + '/') and k.endswith('/in')})


def test_fixture_inventory(arrays):
  assert len(_cases(arrays, 'uq')) == 7 and len(_cases(arrays, 'nuq')) == 7 and len(_cases(arrays, 'uq_act')) == 3
  assert len(arrays) > 400


def test_uniform_quantize_weights_bit_exact(arrays):
  n = 0
  for c in _cases(arrays, 'uq'):
    w = arrays['uq/%s/in' % c]
    for bits in (1, 2, 4, 8, 16, 32):
      for mode, (use_b, btype, bsize) in UQ_MODES.items():
        ref = arrays['uq/%s/b%d/%s' % (c, bits, mode)]
        got, info = O.uniform_quantize(w, bits, 'weight', use_b, btype, bsize)
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.array_equal(got, ref), (c, bits, mode, float(np.max(np.abs(got - ref))))
        if bits == 8:
          assert info['bucket_storage_bits'] == int(arrays['uq/%s/storage/%s' % (c, mode)]), (c, mode)
        n += 1
  assert n == 7 * 6 * 4


def test_uniform_quantize_activations_bit_exact(arrays):
  for c in _cases(arrays, 'uq_act'):
    a = arrays['uq_act/%s/in' % c]
    for bits in (2, 8, 32):
      got, _ = O.uniform_quantize(a, bits, mode='activation', use_buckets=True, bucket_type='channel')
      assert np.array_equal(got, arrays['uq_act/%s/b%d' % (c, bits)]), (c, bits)


def test_split_bucket_quirk_is_strided(arrays):
  """3x3x5x7 = 315 elements, bucket_size 256 -> 2 buckets of 256 with 197 padded copies of the last
  element; bucket j holds flat[i*2 + j] (a strided set), so the two (alpha, beta) pairs come from
  the even / odd flat indices of the padded vector."""
  w = arrays['uq/conv3x3x5x7/in']
  _, info = O.uniform_quantize(w, 8, 'weight', True, 'split', 256)
  assert info['bucket_num'] == 2 and info['padded_num'] == 197
  flat = np.concatenate([w.reshape(-1), np.full(197, w.reshape(-1)[-1], np.float32)])
  assert np.array_equal(info['beta'], np.array([flat[0::2].min(), flat[1::2].min()], np.float32))


NUQ_MODES = {'tensor': (False, 'split', 0, 'quantile'), 'tensor_uniform': (False, 'split', 0, 'uniform'),
             'channel': (True, 'channel', 0, 'quantile'), 'split64': (True, 'split', 64, 'quantile')}


def test_nonuniform_quantize_and_cluster_init_bit_exact(arrays):
  n = 0
  for c in _cases(arrays, 'nuq'):
    w = arrays['nuq/%s/in' % c]
    for bits in (1, 2, 4):
      for mode, (use_b, btype, bsize, style) in NUQ_MODES.items():
        ref = arrays['nuq/%s/b%d/%s/out' % (c, bits, mode)]
        ref_c = arrays['nuq/%s/b%d/%s/clusters' % (c, bits, mode)]
        got, info = O.nuq_quantize(w, bits, None, use_b, btype, bsize, style)
        assert np.array_equal(info['codebook'], ref_c), (c, bits, mode)
        assert np.array_equal(got, ref), (c, bits, mode, float(np.max(np.abs(got - ref))))
        # a supplied codebook (the trained `clusters` variable) takes the same path
        got2, _ = O.nuq_quantize(w, bits, ref_c, use_b, btype, bsize, style)
        assert np.array_equal(got2, ref)
        n += 1
  assert n == 7 * 3 * 4


def test_quantiser_gradients_match_the_executed_reference_graph_code(arrays):
  """Backward rules of the quantisers.  The reference rewires TF's gradient registry while it builds them
  (gradient_override_map {'Round':'Identity'} / {'Mul':'Add','Sign':'Identity'}, tf.stop_gradient on the ranges);
  tests/golden/make_reference_golden.py executes those very functions over oracle/tf_eager_grad_stub.py (torch autograd
  honouring the override map) and records d(sum(y*G))/dx and d/d(clusters).  The oracle's hand-stated rules
  (uniform_quantize_grad, nuq_backward) must reproduce them; the two stand-ins must agree on the forward values."""
  cases = sorted({k.split('/')[1] for k in arrays if k.startswith("qgrad/") and k.split('/')[1] != 'act'})
  assert cases == ['conv1x1x16x32', 'conv3x3x5x7', 'dense37x5']
  for c in cases:
    w, G = arrays['qgrad/%s/in' % c], arrays['qgrad/%s/upstream' % c]
    gmax = float(np.abs(G).max())
    for mode, use_b, btype, bsize in (('tensor', False, 'split', 0), ('channel', True, 'channel', 0), ('split64', True, 'split', 64)):
      # uniform, 4 bits: straight-through identity (TF computes g * alpha / alpha: 1-2 ulp)
      y, _ = O.uniform_quantize(w, 4, 'weight', use_b, btype, bsize)
      assert np.max(np.abs(y - arrays['qgrad/%s/uq4/%s/out' % (c, mode)])) <= 1e-6
      assert np.max(np.abs(O.uniform_quantize_grad(G) - arrays['qgrad/%s/uq4/%s/dx' % (c, mode)])) <= 4e-7 * gmax
      if 'uq/%s/b4/%s' % (c, mode) in arrays:                # the NumPy stand-in's forward fixture of the same call
        assert np.max(np.abs(arrays['uq/%s/b4/%s' % (c, mode)] - arrays['qgrad/%s/uq4/%s/out' % (c, mode)])) <= 1e-6
      # non-uniform, 3 bits: dx = g, d(clusters)[j, b] = sum over the elements assigned to j of alpha_b * g
      pre = 'qgrad/%s/nuq3/%s/' % (c, mode)
      y, info = O.nuq_quantize(w, 3, None, use_b, btype, bsize, 'quantile')
      assert np.array_equal(info['codebook'].reshape(arrays[pre + 'clusters'].shape), arrays[pre + 'clusters'])
      assert np.max(np.abs(y - arrays[pre + 'out'])) <= 1e-6
      gw, dc = O.nuq_backward(G, info, use_b, btype, bsize)
      assert np.max(np.abs(gw - arrays[pre + 'dx'])) <= 4e-7 * gmax
      ref_dc = arrays[pre + 'dclusters']
      assert dc.shape == ref_dc.shape
      assert np.max(np.abs(dc - ref_dc)) <= 2e-5 * max(1.0, float(np.abs(ref_dc).max())), (c, mode)
  a, Ga = arrays['qgrad/act/in'], arrays['qgrad/act/upstream']
  assert np.max(np.abs(O.uniform_quantize_grad(Ga) - arrays['qgrad/act/dx'])) <= 4e-7 * float(np.abs(Ga).max())


def test_distillation_loss_bit_exact(arrays):
  for name in ('b4c10', 'b3c1001', 'b5c7_T1'):
    T, w = arrays['dst/%s/cfg' % name]
    loss, dz = O.distill_loss(arrays['dst/%s/zs' % name], arrays['dst/%s/zt' % name], T, w)
    assert np.float32(loss) == arrays['dst/%s/loss' % name], name
    # closed-form gradient: w / (B T) * (softmax(zs/T) - softmax(zt/T)), checked by finite differences
    zs = arrays['dst/%s/zs' % name].astype(np.float64)
    zt = arrays['dst/%s/zt' % name].astype(np.float64)

    def f(z):
      ls = z / T - np.log(np.sum(np.exp(z / T - (z / T).max(1, keepdims=True)), 1, keepdims=True)) - (z / T).max(1, keepdims=True)
      p = np.exp(zt / T - (zt / T).max(1, keepdims=True))
      p /= p.sum(1, keepdims=True)
      return w * np.mean(-np.sum(p * ls, axis=1))
    i, j = 1, 3
    e = np.zeros_like(zs)
    e[i, j] = 1e-4
    fd = (f(zs + e) - f(zs - e)) / 2e-4
    assert abs(fd - dz[i, j]) <= 1e-5 * max(1.0, abs(fd))


def test_ws_prune_ratio_schedule_bit_exact(arrays):
  for N, step, r_f, ref in arrays['ws/prune_ratio_dyn']:
    got = O.ws_prune_ratio_dyn(int(step), int(N), r_f)
    assert np.float32(got) == np.float32(ref), (N, step, r_f)


def test_ws_mask_refresh_chain_bit_exact(arrays):
  var0 = arrays['ws/chain/var0']
  mask, bkup = np.ones_like(var0), var0.copy()
  for rnd in range(3):
    N, step, r_f = arrays['ws/chain/r%d/step' % rnd]
    var_in = arrays['ws/chain/r%d/var_in' % rnd]
    r_t = O.ws_prune_ratio_dyn(int(step), int(N), r_f)
    var, bkup, mask, thr = O.ws_mask_refresh(var_in, bkup, mask, r_t)
    assert thr == arrays['ws/chain/r%d/thres' % rnd], rnd
    assert np.array_equal(mask, arrays['ws/chain/r%d/mask' % rnd]), rnd
    assert np.array_equal(bkup, arrays['ws/chain/r%d/bkup' % rnd]), rnd
    assert np.array_equal(var, arrays['ws/chain/r%d/var' % rnd]), rnd
  assert 0.45 < 1 - mask.mean() < 0.55                   # reached the final ratio 0.5 at step >= 0.5 N


def test_percentile_nearest_rank(arrays):
  for n in (1, 2, 10, 11, 1000):
    x = arrays['pct/n%d/in' % n]
    for q, ref in zip(arrays['pct/n%d/q' % n], arrays['pct/n%d/out' % n]):
      assert O.percentile_nearest(x, q) == ref, (n, q)


def test_lrn_rate_schedules(host):
  for r in host['lrn_rate_piecewise']:
    got = O.lrn_rate_piecewise(r['step'], r['batch_size'], r['idxs_epoch'], r['decay_rates'], r['nb_smpls_train'],
                               r['lrn_rate_init'], r['batch_size_norm'], r['nb_epochs_rat'])
    assert np.float32(got) == np.float32(r['lrn_rate']), r
  for r in host['setup_bnds_decay_rates']:
    fn = O.uq_setup_bnds_decay_rates if r['learner'] == 'uq' else O.nuq_setup_bnds_decay_rates
    init_lr, bnds, decay, steps = fn(r['model'], r['dataset'], r['batch_size'], r['nb_smpls_train'], 1e-1, 256.0, 60,
                                     r['enbl_multi_gpu'], r['mgw_size'], r['enbl_warm_start'])
    assert (init_lr, list(bnds), list(decay), steps) == (r['init_lr'], r['bnds'], r['decay_rates'],
                                                         r['finetune_steps']), r


def test_ws_host_side(host):
  m = host['get_maskable_vars']
  assert O.get_maskable_var_names(m['names']) == m['maskable']
  assert [[n, r] for n, r in O.pr_uniform([n for n, _ in host['pr_uniform']], 0.75)] == host['pr_uniform']
  h = host['pr_heurist']
  got = O.pr_heurist([n for n, _ in h['out']], [int(np.prod(s)) for s in h['shapes']], h['prune_ratio'])
  for (n, r), (n2, r2) in zip(got, h['out']):
    assert n == n2 and abs(r - r2) <= 1e-15


@pytest.mark.parametrize('model,dataset,size,ncls,shape', [
    ('resnet', 'cifar_10', 20, 10, (32, 32, 3)), ('resnet', 'ilsvrc_12', 50, 11, (64, 64, 3)),
    ('resnet', 'ilsvrc_12', 18, 7, (64, 64, 3)), ('lenet', 'cifar_10', 0, 10, (32, 32, 3))])
def test_network_definitions_match_the_reference_code(arrays, host, model, dataset, size, ncls, shape):
  """Logits of the reference's own network code (executed over oracle/tf_stub.py's tf.layers stand-ins) vs the
  restated networks of oracle/learner_oracle.py on the same seeded variables, and the creation order of the
  matmul kernels, which decides the per-layer bit widths (uq utils.py:115-134)."""
  import torch
  from oracle import learner_oracle as LO
  vals, images = LO.net_fixture_recipe(model, dataset, size, ncls, shape)
  key = '%s_%s_%d' % (model, dataset, size)
  x = torch.from_numpy(images).permute(0, 3, 1, 2)
  for mode in (('train', 'eval') if model == 'resnet' else ('eval',)):
    s = LO.Scope(vals, 'model', trainable=False)
    s.training = mode == 'train'
    s._begin()
    with torch.no_grad():
      if model == 'resnet':
        got = LO.resnet_v2_forward(s, x, LO.resnet_cfg(dataset, size)).numpy()
      else:
        got = LO.lenet_forward(s, x, ncls).numpy()
    ref = arrays['net/%s/%s' % (key, mode)]
    assert np.max(np.abs(got - ref)) <= 2e-4 * max(1.0, float(np.max(np.abs(ref)))), (key, mode)
    assert s.matmul_names == host['net_matmul_order'][key], key


@pytest.mark.parametrize('dm,shape,ncls', [(50, (64, 64, 3), 16), (100, (96, 96, 3), 8)])
def test_mobilenet_definition_matches_the_reference_code(arrays, host, dm, shape, ncls):
  import torch
  from oracle import learner_oracle as LO
  vals, images = LO.net_fixture_recipe('mobilenet_v1', 'ilsvrc_12', dm, ncls, shape)
  x = torch.from_numpy(images).permute(0, 3, 1, 2)
  for mode in ('train', 'eval'):
    s = LO.Scope(vals, 'model', trainable=False)
    s.training = mode == 'train'
    s._begin()
    with torch.no_grad():
      got = LO.mobilenet_v1_forward(s, x, {}).numpy()
    ref = arrays['net/mobilenet_v1_%d/%s' % (dm, mode)]
    assert np.max(np.abs(got - ref)) <= 2e-4 * max(1.0, float(np.max(np.abs(ref)))), (dm, mode)
    assert s.matmul_names == host['net_matmul_order']['mobilenet_v1_%d' % dm]


@pytest.mark.parametrize('key,bn_filter,ilsvrc', [('resnet_ilsvrc12', True, True), ('resnet_cifar10', True, False),
                                                  ('mobilenet_ilsvrc12', True, True), ('lenet_cifar10', False, False)])
def test_model_loss_and_metrics_match_the_reference_code(arrays, key, bn_filter, ilsvrc):
  """ModelHelper.calc_loss of the reference, executed: CE + loss_w_dcy * sum l2_loss(v) over the trainables whose
  name lacks 'batch_normalization' (slim's `BatchNorm/...` names therefore ARE regularised), top-k
  metrics with ties counted in favour, 'accuracy' == top-5 on ILSVRC-12."""
  names = [k[len('loss/var/'):].replace('|', '/') for k in arrays if k.startswith('loss/var/')]
  l2_vars = [arrays['loss/var/' + n.replace('/', '|')] for n in names if not (bn_filter and 'batch_normalization' in n)]
  loss, _, _ = O.model_loss(arrays['loss/%s/labels' % key], arrays['loss/%s/logits' % key], l2_vars, 3e-3)
  ref = arrays['loss/%s/loss' % key]
  assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref)), (float(loss), float(ref))
  m = (O.metrics_ilsvrc if ilsvrc else O.metrics_cifar)(arrays['loss/%s/labels' % key], arrays['loss/%s/logits' % key])
  got_keys = sorted(k.split('/')[-1] for k in arrays if k.startswith('loss/%s/metric/' % key))
  assert got_keys == sorted(m.keys())
  for k in got_keys:
    assert float(m[k]) == float(arrays['loss/%s/metric/%s' % (key, k)]), k


def test_cpg_proximal_step_is_the_reference_statements():
  """oracle cpg_proximal_step against tests/golden/reference_cpg.npz: the five statements of the reference's `prune` layer op
  (learners/channel_pruning_gpu/learner.py:379-383) located with `ast` and executed over oracle/tf_stub.py
  (tests/golden/make_reference_cpg_golden.py).  Bit for bit: new kernel, per-input-channel norms, threshold.  The stub's
  reduce_sum is NumPy's, so the summation ORDER inside the norm is the stub's, not TensorFlow's; everything else (which axes,
  which tensor the percentile sees and its 'nearest' index, the shrink expression, one float32 rounding per op) is the
  reference's code.  The cases hold the two ends (q = 0: one channel zeroed; q = 100: all), a fractional 'nearest' index and
  a half-to-even one."""
  with np.load(os.path.join(GOLD, 'reference_cpg.npz')) as z:
    a = {k: z[k] for k in z.files}
  with open(os.path.join(GOLD, 'reference_cpg.json')) as f:
    meta = json.load(f)
  assert meta['reference'] == 'learners/channel_pruning_gpu/learner.py:379-383' and len(meta['cases']) == 9
  for i, case in enumerate(meta['cases']):
    k = 'cpg/%d/' % i
    new, norm, thr = O.cpg_proximal_step(a[k + 'w'], a[k + 'g'], case['lrn_rate'], case['prune_perctl'])
    assert new.dtype == np.float32 and new.shape == tuple(case['shape'])
    assert np.array_equal(norm, a[k + 'norm']), (i, 'norm')
    assert np.float32(thr) == a[k + 'thr'], (i, 'threshold')
    assert np.array_equal(new, a[k + 'new']), (i, float(np.max(np.abs(new - a[k + 'new']))))
    assert int(np.sum(np.all(new == 0, axis=(0, 1, 3)))) == case['channels_zeroed']
  zeroed = [c['channels_zeroed'] for c in meta['cases']]
  assert zeroed[1] == 1 and zeroed[2] == meta['cases'][2]['shape'][2]
