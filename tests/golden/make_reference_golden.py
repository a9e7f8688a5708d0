#!/usr/bin/env python
"""Generate tests/golden/reference_*.npz by EXECUTING THE REFERENCE'S OWN hot-path functions.

Runs only in the build container (it reads /root/reference); the fixtures it writes are committed
and are what the tests use -- nothing in tests/ reads /root/reference.

How: TensorFlow 1.x is not installable here, so the functions are lifted out of the reference files
with `ast` (class / function definitions are compiled in place; no reference source is copied into
this repository) and executed with `tf` bound to oracle/tf_stub.py, a NumPy-eager stand-in for the
~40 TF ops they call (see that file for exactly what remains a restatement of TF semantics).

    python tests/golden/make_reference_golden.py            # rewrites tests/golden/reference_*.npz

Lifted (paths relative to /root/reference):
  learners/uniform_quantization/utils.py      class UniformQuantization    (__uniform_quantize, __scale,
                                              __inv_scale, __split_bucket, __channel_bucket)
  learners/nonuniform_quantization/utils.py   class NonUniformQuantization (__nonuni_quantize,
                                              __bucket_quantize, __quantile_init, __uniform_init,
                                              __build_[bucket_]norm_quant_point, __uniform_quantize)
  learners/distillation_helper.py             DistillationHelper.calc_loss
  learners/weight_sparsification/learner.py   WeightSparseLearner.__calc_prune_ratio_dyn + the
                                              prune_op chain of __build_masks (:283-288, re-enacted
                                              op by op below with the same stub ops)
  learners/weight_sparsification/utils.py     get_maskable_vars
  learners/weight_sparsification/pr_optimizer.py  __calc_uniform/heurist_prune_ratios
  learners/uniform_quantization/learner.py    setup_bnds_decay_rates
  learners/nonuniform_quantization/learner.py setup_bnds_decay_rates
  utils/lrn_rate_utils.py                     setup_lrn_rate_piecewise_constant / _exponential_decay
  learners/channel_pruning/channel_pruner.py  ChannelPruner.compute_pruned_kernel + featuremap_reconstruction
                                              (NumPy + the real scikit-learn LassoLars / LinearRegression)
  utils/external/resnet_model.py              the WHOLE module (Model.__call__, blocks, fixed padding, BN constants)
  nets/lenet_at_cifar10.py                    forward_fn
  utils/external/mobilenet_v1.py              the WHOLE module over tf.contrib.slim stand-ins
  nets/{resnet_at_ilsvrc12,resnet_at_cifar10,mobilenet_at_ilsvrc12,lenet_at_cifar10}.py   ModelHelper.calc_loss
  utils/get_path_args.py                      run as a script (pure Python)
"""
import ast
import json
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import tf_stub  # noqa: E402

tf = tf_stub.install()
FLAGS = tf.app.flags.FLAGS


def lift(path, names, extra_globals=None):
  """Compile the top-level definitions `names` (classes / functions; 'Class.method' for a method) of a
  reference file into a fresh namespace with `tf` = the stub."""
  src = open(os.path.join(REF, path)).read()
  tree = ast.parse(src)
  ns = {'tf': tf, 'np': np, 'FLAGS': FLAGS, '__name__': 'lifted'}
  ns.update(extra_globals or {})
  body, found, classes = [], 0, {}
  for want in names:
    if '.' in want:
      cls, meth = want.split('.')
      for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
          for sub in node.body:
            if isinstance(sub, ast.FunctionDef) and sub.name == meth:
              sub.decorator_list = []
              if cls not in classes:           # a shell class of the same name keeps the name mangling
                classes[cls] = ast.ClassDef(name=cls, bases=[], keywords=[], body=[], decorator_list=[])
                body.append(classes[cls])
              classes[cls].body.append(sub)
              found += 1
    else:
      for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name == want:
          body.append(node)
          found += 1
  assert found == len(names), (path, names)
  mod = ast.Module(body=body, type_ignores=[])
  ast.fix_missing_locations(mod)
  exec(compile(mod, os.path.join(REF, path), 'exec'), ns)
  return ns


def t32(a):
  return tf_stub.T(np.asarray(a, dtype=np.float32))


def bits_t(b):
  return tf_stub.T(np.int64(b))


# -------------------------------------------------------------------------------------------------
def weight_cases(rng):
  cases = {
      'conv3x3x5x7': rng.randn(3, 3, 5, 7).astype(np.float32) * 0.1,          # 315 elems: split pad 197 @256
      'conv1x1x16x32': rng.randn(1, 1, 16, 32).astype(np.float32) * 0.05,     # 512 elems: exact split
      'dense20x10': rng.uniform(-1, 1, (20, 10)).astype(np.float32),
      'depthwise3x3x8x1': rng.randn(3, 3, 8, 1).astype(np.float32),
      'ties8': np.array([0.0, 0.5, 1.5, 2.5, 3.5, 4.5, 6.5, 7.0], np.float32).reshape(1, 1, 2, 4),
      'constant': np.full((1, 1, 4, 4), 0.25, np.float32),
      'tiny_range': (1.0 + rng.randn(2, 2, 3, 3) * 1e-7).astype(np.float32),
  }
  return cases


def gen_uniform(out):
  ns = lift('learners/uniform_quantization/utils.py', ['UniformQuantization'], {'ge': tf.contrib.graph_editor})
  UQ = ns['UniformQuantization']
  rng = np.random.RandomState(20240601)
  cases = weight_cases(rng)
  for cname, w in cases.items():
    out['uq/%s/in' % cname] = w
    for bits in (1, 2, 4, 8, 16, 32):
      for mode_name, use_b, btype, bsize in (('tensor', False, 'split', 0), ('channel', True, 'channel', 0),
                                             ('split256', True, 'split', 256), ('split4', True, 'split', 4)):
        q = UQ(tf.Session(), bsize, use_b, btype)
        y = q._UniformQuantization__uniform_quantize(t32(w), bits_t(bits), 'weight', 'm/c')
        out['uq/%s/b%d/%s' % (cname, bits, mode_name)] = y.a.astype(np.float32)
        if bits == 8:
          out['uq/%s/storage/%s' % (cname, mode_name)] = np.int64(int(np.asarray(tf_stub._raw(q.bucket_storage))))
  # activations: per-tensor over the whole batch, also when use_buckets is on
  acts = {'relu_2x4x4x3': np.maximum(rng.randn(2, 4, 4, 3), 0).astype(np.float32),
          'relu6_big': np.minimum(np.maximum(rng.randn(4, 8, 8, 16) * 4, 0), 6).astype(np.float32),
          'all_zero': np.zeros((2, 2, 2, 2), np.float32)}
  for aname, a in acts.items():
    out['uq_act/%s/in' % aname] = a
    for bits in (2, 8, 32):
      q = UQ(tf.Session(), 256, True, 'channel')
      out['uq_act/%s/b%d' % (aname, bits)] = q._UniformQuantization__uniform_quantize(
          t32(a), bits_t(bits), 'activation', 'm/a').a.astype(np.float32)


def gen_nonuniform(out):
  ns = lift('learners/nonuniform_quantization/utils.py', ['NonUniformQuantization'], {'ge': tf.contrib.graph_editor})
  NQ = ns['NonUniformQuantization']
  rng = np.random.RandomState(20240602)
  cases = weight_cases(rng)
  del cases['constant']                     # alpha = 1e-10: x_hat == 0 everywhere, covered by 'ties'
  cases['conv3x3x16x24'] = rng.randn(3, 3, 16, 24).astype(np.float32) * 0.2
  for cname, w in cases.items():
    out['nuq/%s/in' % cname] = w
    for bits in (1, 2, 4):
      for mode_name, use_b, btype, bsize, style in (('tensor', False, 'split', 0, 'quantile'),
                                                    ('tensor_uniform', False, 'split', 0, 'uniform'),
                                                    ('channel', True, 'channel', 0, 'quantile'),
                                                    ('split64', True, 'split', 64, 'quantile')):
        tf_stub.created_variables.clear()
        q = NQ(tf.Session(), bsize, use_b, style, btype)
        y = q.quant_fn(t32(w), bits_t(bits), 'weight', 'm/c')
        (cb,) = list(tf_stub.created_variables.values())
        out['nuq/%s/b%d/%s/out' % (cname, bits, mode_name)] = y.a.astype(np.float32)
        out['nuq/%s/b%d/%s/clusters' % (cname, bits, mode_name)] = cb.a.astype(np.float32)


def gen_quantiser_gradients(out):
  """The quantisers' BACKWARD rules, by executing the same reference functions over oracle/tf_eager_grad_stub.py (torch
  autograd honouring gradient_override_map / stop_gradient): d(sum(y * G))/dx and, for NUQ, d/d(clusters)."""
  sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
  import torch
  import oracle.tf_eager_grad_stub as tg
  rng = np.random.RandomState(20240701)
  cases = weight_cases(np.random.RandomState(20240601))
  cases = {k: cases[k] for k in ('conv3x3x5x7', 'conv1x1x16x32')}
  cases['dense37x5'] = (rng.randn(37, 5) * 0.3).astype(np.float32)
  UQ = lift('learners/uniform_quantization/utils.py', ['UniformQuantization'], {'tf': tg, 'ge': None})['UniformQuantization']
  NQ = lift('learners/nonuniform_quantization/utils.py', ['NonUniformQuantization'], {'tf': tg, 'ge': None})['NonUniformQuantization']
  for cname, w in cases.items():
    G = rng.randn(*w.shape).astype(np.float32)
    out['qgrad/%s/in' % cname], out['qgrad/%s/upstream' % cname] = w, G
    for mode_name, use_b, btype, bsize in (('tensor', False, 'split', 0), ('channel', True, 'channel', 0), ('split64', True, 'split', 64)):
      x = tg.T(torch.tensor(w, requires_grad=True))
      q = UQ(tg.Session(), bsize, use_b, btype)
      y = q._UniformQuantization__uniform_quantize(x, tg.T(np.int64(4)), 'weight', 'm/c')
      (dx,) = tg.gradients(tg.reduce_sum(y * tg.T(G)), [x])
      out['qgrad/%s/uq4/%s/out' % (cname, mode_name)] = y.numpy().astype(np.float32)
      out['qgrad/%s/uq4/%s/dx' % (cname, mode_name)] = dx.numpy().astype(np.float32)
      tg.created_variables.clear()
      x = tg.T(torch.tensor(w, requires_grad=True))
      q = NQ(tg.Session(), bsize, use_b, 'quantile', btype)
      y = q.quant_fn(x, tg.T(np.int64(3)), 'weight', 'm/c')
      (cb,) = list(tg.created_variables.values())
      dx, dc = tg.gradients(tg.reduce_sum(y * tg.T(G)), [x, cb])
      out['qgrad/%s/nuq3/%s/out' % (cname, mode_name)] = y.numpy().astype(np.float32)
      out['qgrad/%s/nuq3/%s/clusters' % (cname, mode_name)] = cb.numpy().astype(np.float32)
      out['qgrad/%s/nuq3/%s/dx' % (cname, mode_name)] = dx.numpy().astype(np.float32)
      out['qgrad/%s/nuq3/%s/dclusters' % (cname, mode_name)] = dc.numpy().astype(np.float32)
  # activation quantiser (per-tensor range under stop_gradient, Round -> Identity): gradient = identity
  a = np.maximum(rng.randn(2, 4, 4, 3), 0).astype(np.float32)
  Ga = rng.randn(*a.shape).astype(np.float32)
  x = tg.T(torch.tensor(a, requires_grad=True))
  y = UQ(tg.Session(), 256, True, 'channel')._UniformQuantization__uniform_quantize(x, tg.T(np.int64(8)), 'activation', 'm/a')
  (dx,) = tg.gradients(tg.reduce_sum(y * tg.T(Ga)), [x])
  out['qgrad/act/in'], out['qgrad/act/upstream'], out['qgrad/act/dx'] = a, Ga, dx.numpy().astype(np.float32)


def gen_distill(out):
  ns = lift('learners/distillation_helper.py', ['DistillationHelper.calc_loss'])
  calc_loss = ns['DistillationHelper'].calc_loss
  rng = np.random.RandomState(20240603)
  for name, (B, C, T, w) in {'b4c10': (4, 10, 4.0, 4.0), 'b3c1001': (3, 1001, 4.0, 4.0), 'b5c7_T1': (5, 7, 1.0, 0.5)}.items():
    FLAGS.tempr_dst, FLAGS.loss_w_dst = T, w
    zs = (rng.randn(B, C) * 3).astype(np.float32)
    zt = (rng.randn(B, C) * 3).astype(np.float32)
    loss = calc_loss(None, t32(zs), t32(zt))
    out['dst/%s/zs' % name], out['dst/%s/zt' % name] = zs, zt
    out['dst/%s/cfg' % name] = np.array([T, w], np.float64)
    out['dst/%s/loss' % name] = np.float32(loss.a)


def gen_ws(out):
  ns = lift('learners/weight_sparsification/learner.py', ['WeightSparseLearner.__calc_prune_ratio_dyn'])
  cls = ns['WeightSparseLearner']
  FLAGS.ws_iter_ratio_beg, FLAGS.ws_iter_ratio_end, FLAGS.ws_prune_ratio_exp = 0.1, 0.5, 3.0
  rows = []
  for N in (20, 1000, 97656):
    for step in sorted(set([0, 1, N // 10 - 1, N // 10, N // 10 + 1, N // 4, N // 3, N // 2 - 1, N // 2, N // 2 + 1, N - 1])):
      for r_f in (0.5, 0.75, 0.9):
        self = types.SimpleNamespace(nb_iters_train=N, global_step=tf_stub.T(np.int64(step)))
        r = cls._WeightSparseLearner__calc_prune_ratio_dyn(self, r_f)
        rows.append((N, step, r_f, float(np.float32(r.a))))
  out['ws/prune_ratio_dyn'] = np.array(rows, np.float64)
  # the prune_op chain of __build_masks (ws learner.py:283-288), op by op with the stub ops, two rounds
  rng = np.random.RandomState(20240604)
  var = (rng.randn(3, 3, 8, 16) * 0.1).astype(np.float32)
  var.reshape(-1)[:7] = var.reshape(-1)[7:14]                  # ties in |w|
  mask = np.ones_like(var)
  bkup = var.copy()
  out['ws/chain/var0'] = var
  for rnd, (N, step, r_f, drift) in enumerate([(1000, 200, 0.5, 0.0), (1000, 400, 0.5, 0.03), (1000, 500, 0.5, 0.03)]):
    var = (var + (rng.randn(*var.shape) * drift).astype(np.float32) * mask).astype(np.float32)   # training moves kept weights
    self = types.SimpleNamespace(nb_iters_train=N, global_step=tf_stub.T(np.int64(step)))
    prune_ratio = cls._WeightSparseLearner__calc_prune_ratio_dyn(self, r_f)
    out['ws/chain/r%d/var_in' % rnd] = var
    v, b, m = t32(var), t32(bkup), t32(mask)
    b = tf.where(tf_stub.T(m.a > 0.5), v, b)
    thres = tf.contrib.distributions.percentile(tf.abs(b), prune_ratio * 100)
    m = tf.cast(tf_stub.T(tf.abs(b).a > thres.a), tf.float32)
    v = b * m
    var, bkup, mask = v.a, b.a, m.a
    out['ws/chain/r%d/step' % rnd] = np.array([N, step, r_f], np.float64)
    out['ws/chain/r%d/var' % rnd], out['ws/chain/r%d/bkup' % rnd] = var, bkup
    out['ws/chain/r%d/mask' % rnd], out['ws/chain/r%d/thres' % rnd] = mask, np.float32(thres.a)
  # percentile, nearest rank, incl. the tiny lengths of the KAT list
  for n in (1, 2, 10, 11, 1000):
    x = rng.randn(n).astype(np.float32)
    out['pct/n%d/in' % n] = x
    qs = np.array([0.0, 12.5, 33.333, 50.0, 75.0, 99.9, 100.0])
    out['pct/n%d/q' % n] = qs
    out['pct/n%d/out' % n] = np.array([tf.contrib.distributions.percentile(t32(x), q).a for q in qs], np.float32)


def gen_schedules(meta):
  ns = lift('utils/lrn_rate_utils.py', ['setup_lrn_rate_piecewise_constant', 'setup_lrn_rate_exponential_decay'])
  rows = []
  for (nb_smpls, lr0, bsn, rat, bs, idxs, rates) in [
      (50000, 1e-1, 128, 1.0, 128, [100, 150, 200], [1.0, 0.1, 0.01, 0.001]),
      (50000, 1e-2, 128, 0.004, 16, [100, 150, 200], [1.0, 0.1, 0.01, 0.001]),
      (1281167, 1e-1, 256, 1.0, 256 * 8, [30, 60, 80, 90], [1.0, 0.1, 0.01, 0.001, 0.0001])]:
    FLAGS.nb_smpls_train, FLAGS.lrn_rate_init, FLAGS.batch_size_norm, FLAGS.nb_epochs_rat = nb_smpls, lr0, bsn, rat
    per = float(nb_smpls) / bs
    steps = sorted(set([0, 1] + [int(per * e * rat) + d for e in idxs for d in (-1, 0, 1)] + [10 ** 7]))
    for st in steps:
      if st < 0:
        continue
      lr = ns['setup_lrn_rate_piecewise_constant'](tf_stub.T(np.int64(st)), bs, list(idxs), list(rates))
      rows.append(dict(nb_smpls_train=nb_smpls, lrn_rate_init=lr0, batch_size_norm=bsn, nb_epochs_rat=rat,
                       batch_size=bs, idxs_epoch=idxs, decay_rates=rates, step=st, lrn_rate=float(lr.a)))
  meta['lrn_rate_piecewise'] = rows

  class _Mgw(object):
    n = 1
    @classmethod
    def size(cls):
      return cls.n
  rows = []
  for path, key, epochs_flag in (('learners/uniform_quantization/learner.py', 'uq', 'uql_quant_epochs'),
                                 ('learners/nonuniform_quantization/learner.py', 'nuq', 'nuql_quant_epochs')):
    ns = lift(path, ['setup_bnds_decay_rates'], {'mgw': _Mgw})
    for model, dataset, nb_smpls, bs in (('resnet_20', 'cifar_10', 50000, 128), ('resnet_50', 'ilsvrc_12', 1281167, 64),
                                         ('mobilenet_v1', 'ilsvrc_12', 1281167, 64)):
      for multi, n, warm in ((False, 1, False), (True, 8, False), (True, 8, True)):
        _Mgw.n = n
        FLAGS.batch_size, FLAGS.enbl_multi_gpu, FLAGS.nb_smpls_train = bs, multi, nb_smpls
        FLAGS.lrn_rate_init, FLAGS.batch_size_norm, FLAGS.enbl_warm_start = 1e-1, 256.0, warm
        setattr(FLAGS, epochs_flag, 60)
        init_lr, bnds, decay, steps = ns['setup_bnds_decay_rates'](model, dataset)
        rows.append(dict(learner=key, model=model, dataset=dataset, nb_smpls_train=nb_smpls, batch_size=bs,
                         enbl_multi_gpu=multi, mgw_size=n, enbl_warm_start=warm, init_lr=init_lr, bnds=bnds,
                         decay_rates=decay, finetune_steps=steps))
  meta['setup_bnds_decay_rates'] = rows


def gen_ws_host(meta):
  ns = lift('learners/weight_sparsification/utils.py', ['get_maskable_vars'])
  names = ['model/resnet_model/conv2d/kernel:0', 'model/resnet_model/batch_normalization/gamma:0',
           'model/resnet_model/dense/kernel:0', 'model/resnet_model/dense/bias:0',
           'model/MobilenetV1/Conv2d_0/weights:0', 'model/MobilenetV1/Conv2d_1_depthwise/depthwise_weights:0',
           'model/MobilenetV1/Conv2d_1_pointwise/weights:0', 'model/MobilenetV1/Logits/Conv2d_1c_1x1/weights:0',
           'model/MobilenetV1/Logits/Conv2d_1c_1x1/biases:0']
  vs = [types.SimpleNamespace(name=n) for n in names]
  meta['get_maskable_vars'] = {'names': names, 'maskable': [v.name for v in ns['get_maskable_vars'](vs)]}
  ns = lift('learners/weight_sparsification/pr_optimizer.py',
            ['PROptimizer.__calc_uniform_prune_ratios', 'PROptimizer.__calc_heurist_prune_ratios'])
  cls = ns['PROptimizer']
  shapes = [(3, 3, 16, 16), (3, 3, 16, 32), (1, 1, 16, 32), (64, 10)]
  mv = [types.SimpleNamespace(name='v%d:0' % i) for i in range(len(shapes))]
  FLAGS.ws_prune_ratio = 0.75
  tf.shape = lambda v: v
  sess = types.SimpleNamespace(run=lambda v: np.array(shapes[mv.index(v)]))
  self = types.SimpleNamespace(vars_full={'maskable': mv}, sess=sess)
  meta['pr_uniform'] = [[n, float(r)] for n, r in cls._PROptimizer__calc_uniform_prune_ratios(self)]
  meta['pr_heurist'] = {'shapes': shapes, 'prune_ratio': 0.75,
                        'out': [[n, float(r)] for n, r in cls._PROptimizer__calc_heurist_prune_ratios(self)]}


def gen_channel_pruner(out):
  """compute_pruned_kernel + featuremap_reconstruction of the reference's ChannelPruner (channel_pruner.py:
  443-577) executed as they are (NumPy + the real scikit-learn); np.random is seeded because the reference
  samples with the global generator."""
  from timeit import default_timer as timer
  from sklearn.linear_model import LassoLars, LinearRegression
  ns = lift('learners/channel_pruning/channel_pruner.py',
            ['ChannelPruner.compute_pruned_kernel', 'ChannelPruner.featuremap_reconstruction'],
            {'LassoLars': LassoLars, 'LinearRegression': LinearRegression, 'timer': timer})
  cls = ns['ChannelPruner']
  cls.featuremap_reconstruction = classmethod(cls.featuremap_reconstruction)
  FLAGS.debug, FLAGS.cp_quadruple = False, False
  for name, (seed, n, kh, cin, cout, rank, c_new) in {'pw24': (1, 800, 1, 24, 16, 8, 12), 'k3c16': (2, 800, 3, 16, 12, 6, 8),
                                                       'pw32': (3, 1200, 1, 32, 20, 10, 10)}.items():
    rng = np.random.RandomState(seed)
    basis = rng.randn(n, kh, kh, rank)
    X = np.einsum('nhwr,rc->nhwc', basis, rng.randn(rank, cin)) + 0.05 * rng.randn(n, kh, kh, cin)
    W2 = rng.randn(kh, kh, cin, cout) * 0.2
    Y = X.reshape(n, -1) @ W2.reshape(-1, cout)
    np.random.seed(77)
    self = cls.__new__(cls)
    idxs, newW2 = cls.compute_pruned_kernel(self, X, W2, Y, c_new=c_new)
    # the inputs are regenerated by the test from this recipe (same NumPy Mersenne stream)
    out['cp/%s/recipe' % name] = np.array([seed, n, kh, cin, cout, rank, c_new], np.int64)
    out['cp/%s/idxs' % name], out['cp/%s/newW2' % name] = np.asarray(idxs, bool), np.asarray(newW2)


NET_CASES = [('resnet', 'cifar_10', 20, 10, (32, 32, 3)), ('resnet', 'ilsvrc_12', 50, 11, (64, 64, 3)),
             ('resnet', 'ilsvrc_12', 18, 7, (64, 64, 3)), ('lenet', 'cifar_10', 0, 10, (32, 32, 3))]


MOBILENET_CASES = [(50, (64, 64, 3), 16), (100, (96, 96, 3), 8)]      # (depth multiplier in %, input, classes)


def gen_networks(out, meta):
  """Execute the reference's own NETWORK DEFINITIONS (utils/external/resnet_model.py as a whole module,
  nets/lenet_at_cifar10.py:forward_fn) over the stub's tf.layers stand-ins on seeded variables; store the logits
  and the order in which the code asked for its variables (= TF's creation order)."""
  import importlib.util
  from oracle.learner_oracle import net_fixture_recipe, resnet_cfg
  spec = importlib.util.spec_from_file_location('ref_resnet_model', os.path.join(REF, 'utils/external/resnet_model.py'))
  resnet = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(resnet)
  lenet = lift('nets/lenet_at_cifar10.py', ['forward_fn'])
  orders = {}
  for model, ds, size, ncls, shape in NET_CASES:
    vals, images = net_fixture_recipe(model, ds, size, ncls, shape)
    key = '%s_%s_%d' % (model, ds, size)
    for training in ((True, False) if model == 'resnet' else (False,)):
      tf_stub.reset_layers(vals)
      with tf.variable_scope('model'):
        if model == 'resnet':
          cfg = resnet_cfg(ds, size)
          net = resnet.Model(size, cfg['bottleneck'], ncls, cfg['num_filters'], cfg['kernel_size'], cfg['conv_stride'],
                             cfg['first_pool_size'], cfg['first_pool_stride'], cfg['block_sizes'], cfg['block_strides'],
                             data_format='channels_last')
          logits = net(tf_stub.T(images), training)
        else:
          FLAGS.nb_classes = ncls
          logits = lenet['forward_fn'](tf_stub.T(images), 'channels_last')
      out['net/%s/%s' % (key, 'train' if training else 'eval')] = np.asarray(logits.a, np.float32)
    orders[key] = [u for u in tf_stub.variables_used if u.endswith('kernel')]
  # MobileNet-v1: the whole slim module (mobilenet_v1, mobilenet_v1_base, mobilenet_v1_arg_scope) over the slim stand-ins;
  # dropout_keep_prob = 1.0 in the training-mode fixture keeps it deterministic for every party
  spec = importlib.util.spec_from_file_location('ref_mobilenet_v1', os.path.join(REF, 'utils/external/mobilenet_v1.py'))
  mbv1 = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mbv1)
  slim = tf.contrib.slim
  for dm, shape, ncls in MOBILENET_CASES:
    vals, images = net_fixture_recipe('mobilenet_v1', 'ilsvrc_12', dm, ncls, shape)
    key = 'mobilenet_v1_%d' % dm
    for training in (True, False):
      tf_stub.reset_layers(vals)
      with tf.variable_scope('model'):
        with slim.arg_scope(mbv1.mobilenet_v1_arg_scope(is_training=training)):
          logits, _ = mbv1.mobilenet_v1(tf_stub.T(images), is_training=training, num_classes=ncls,
                                        depth_multiplier=dm / 100.0, dropout_keep_prob=1.0)
      out['net/%s/%s' % (key, 'train' if training else 'eval')] = np.asarray(logits.a, np.float32)
    orders[key] = [u for u in tf_stub.variables_used if u.endswith('weights')]
  meta['net_matmul_order'] = orders


def gen_model_losses(out):
  """ModelHelper.calc_loss of the four configured nets (CE + loss_w_dcy * L2 over the filtered trainables, metrics)."""
  rng = np.random.RandomState(20240605)
  names = ['model/resnet_model/conv2d/kernel', 'model/resnet_model/batch_normalization/gamma',
           'model/resnet_model/batch_normalization/beta', 'model/resnet_model/dense/kernel', 'model/resnet_model/dense/bias',
           'model/MobilenetV1/Conv2d_0/BatchNorm/gamma', 'model/MobilenetV1/Conv2d_1_depthwise/depthwise_weights']
  shapes = [(3, 3, 4, 8), (8,), (8,), (8, 5), (5,), (8,), (3, 3, 8, 1)]
  tvars = []
  for n, sh in zip(names, shapes):
    v = tf_stub.T((rng.randn(*sh) * 0.5).astype(np.float32), name=n + ':0')
    out['loss/var/%s' % n.replace('/', '|')] = v.a
    tvars.append(v)
  for key, path, ncls in (('resnet_ilsvrc12', 'nets/resnet_at_ilsvrc12.py', 12), ('resnet_cifar10', 'nets/resnet_at_cifar10.py', 10),
                          ('mobilenet_ilsvrc12', 'nets/mobilenet_at_ilsvrc12.py', 12), ('lenet_cifar10', 'nets/lenet_at_cifar10.py', 10)):
    ns = lift(path, ['ModelHelper.calc_loss'])
    FLAGS.loss_w_dcy = 3e-3
    B = 16
    logits = (rng.randn(B, ncls) * 2).astype(np.float32)
    logits[0, :3] = logits[0, 0]                                  # ties for in_top_k
    labels = np.eye(ncls, dtype=np.float32)[rng.randint(0, ncls, B)]
    labels[0] = np.eye(ncls, dtype=np.float32)[1]
    loss, metrics = ns['ModelHelper'].calc_loss(None, tf_stub.T(labels), tf_stub.T(logits), tvars)
    out['loss/%s/logits' % key], out['loss/%s/labels' % key] = logits, labels
    out['loss/%s/loss' % key] = np.float32(loss.a)
    for mk, mv in metrics.items():
      out['loss/%s/metric/%s' % (key, mk)] = np.float32(mv.a)


def gen_path_args(meta):
  conf = os.path.join(HERE, 'path.conf.sample')
  rows = []
  for mode, run in (('local', 'nets/resnet_at_cifar10_run.py'), ('local', 'nets/mobilenet_at_ilsvrc12_run.py'),
                    ('docker', 'nets/resnet_at_ilsvrc12_run.py'), ('seven', 'nets/lenet_at_cifar10_run.py')):
    outp = subprocess.run([sys.executable, os.path.join(REF, 'utils/get_path_args.py'), mode, run, conf],
                          capture_output=True, text=True, cwd=REF)
    rows.append({'mode': mode, 'run': run, 'stdout': outp.stdout.strip(), 'rc': outp.returncode})
  meta['get_path_args'] = rows


def main():
  arrays, meta = {}, {}
  import builtins
  real_print = builtins.print
  builtins.print = lambda *a, **k: None      # the reference prints "Quantized: <scope>" per op
  gen_uniform(arrays)
  gen_nonuniform(arrays)
  gen_quantiser_gradients(arrays)
  gen_distill(arrays)
  gen_ws(arrays)
  gen_channel_pruner(arrays)
  gen_networks(arrays, meta)
  gen_model_losses(arrays)
  gen_schedules(meta)
  gen_ws_host(meta)
  gen_path_args(meta)
  builtins.print = real_print
  np.savez_compressed(os.path.join(HERE, 'reference_arrays.npz'), **arrays)
  with open(os.path.join(HERE, 'reference_host.json'), 'w') as f:
    json.dump(meta, f, indent=1, sort_keys=True)
  print('wrote %d arrays (%.1f KiB) and %d host tables' % (
      len(arrays), os.path.getsize(os.path.join(HERE, 'reference_arrays.npz')) / 1024.0, len(meta)))


if __name__ == '__main__':
  main()
