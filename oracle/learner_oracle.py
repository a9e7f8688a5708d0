"""CPU oracle of a whole learner step  --  TEST INFRASTRUCTURE ONLY (see oracle/pf_oracle.py).

A float32 CPU restatement of what one `sess.run(ops['train'])` of the reference's learners computes
(SURVEY section 3.2-3.4): data -> teacher forward -> weight fake-quant -> student forward (BN with
batch statistics, activation fake-quant) -> CE + coupled L2 [+ distillation] -> backward with
straight-through estimators -> [mask] -> Adam / Momentum.  Used by
  * tests/test_parity_gpu.py         step-level parity of the HIP learners (weights, masks, loss),
  * __graft_entry__.smoke()          one-step check on cuda:0,
  * bench.py `cpu_baseline`          the reference path timed on the host cores ("port").
Nothing under pocketflow_amd/ imports this file.

The fake-quant / codebook / mask / loss / optimiser arithmetic is oracle/pf_oracle.py (NumPy, one
rounding per TF op).  The dense contractions (conv2d, matmul), pooling and the BN reductions are
torch-CPU float32 ops with torch autograd providing their gradients: these are the third-party
primitives of the reference (TF/Eigen kernels), not part of what the HIP library re-implements,
and any float32 implementation of them differs from TF's only in summation order.

Variables live in the REFERENCE layout and naming (HWIO conv kernels, [in, out] dense kernels,
`model/resnet_model/conv2d_3/kernel`, ...), i.e. exactly what VarStore.export_numpy() produces and
what a TF checkpoint of the reference holds.

Network definitions restated here (creation order matters: it defines which matmul / activation op
gets which bit width, uq utils.py:115-134):
  lenet      nets/lenet_at_cifar10.py:34-68
  resnet v2  utils/external/resnet_model.py:55-103 (BN, fixed padding), :156-199 / :257-314 (blocks),
             :489-554 (model)
  mobilenet  utils/external/mobilenet_v1.py:124-139, 168-392, 428-477
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from oracle import pf_oracle as O


# =================================================================================================
# autograd wrappers around the NumPy oracle ops
# =================================================================================================

class _WeightUQ(torch.autograd.Function):
  """__uniform_quantize on a weight (uq utils.py:163-199); backward = identity (Round->Identity)."""

  @staticmethod
  def forward(ctx, w, bits, use_buckets, bucket_type, bucket_size):
    q, _ = O.uniform_quantize(w.detach().numpy(), bits, 'weight', use_buckets, bucket_type, bucket_size)
    return torch.from_numpy(np.ascontiguousarray(q))

  @staticmethod
  def backward(ctx, g):
    return g, None, None, None, None


class _ActUQ(torch.autograd.Function):
  """insert_quant_op_for_activations (uq utils.py:51-79): act(u) -> per-tensor fake-quant."""

  @staticmethod
  def forward(ctx, u, bits, act):
    un = u.detach().numpy()
    q, _ = O.activation_quantize(un, bits, act)
    ctx.save_for_backward(u)
    ctx.act = act
    return torch.from_numpy(np.ascontiguousarray(q))

  @staticmethod
  def backward(ctx, g):
    (u,) = ctx.saved_tensors
    return torch.from_numpy(O.activation_quantize_grad(g.numpy(), u.detach().numpy(), ctx.act)), None, None


class _WeightNUQ(torch.autograd.Function):
  """__nonuni_quantize / __bucket_quantize (nuq utils.py:168-243, 284-347).  Gradients under the
  override map {'Mul':'Add','Sign':'Identity'}: identity to the weight, scatter-sum of alpha*g to
  the codebook (pf_oracle.nuq_backward)."""

  @staticmethod
  def forward(ctx, w, codebook, bits, use_buckets, bucket_type, bucket_size):
    q, info = O.nuq_quantize(w.detach().numpy(), bits, codebook.detach().numpy(), use_buckets, bucket_type,
                             bucket_size)
    ctx.info = info
    ctx.cfg = (use_buckets, bucket_type, bucket_size)
    return torch.from_numpy(np.ascontiguousarray(q))

  @staticmethod
  def backward(ctx, g):
    gw, dc = O.nuq_backward(g.numpy(), ctx.info, *ctx.cfg)
    return torch.from_numpy(gw), torch.from_numpy(np.ascontiguousarray(dc)), None, None, None, None


# =================================================================================================
# one model scope
# =================================================================================================

class QuantSpec(object):
  """What the learner's graph rewrite decided: bit widths per matmul / activation op (creation
  order), bucketing, and (NUQ) the codebooks."""

  def __init__(self, kind: str = 'none', w_bits: Optional[Sequence[Optional[int]]] = None,
               a_bits: Optional[Sequence[Optional[int]]] = None, use_buckets: bool = False,
               bucket_type: str = 'channel', bucket_size: int = 256):
    self.kind = kind                         # none | uniform | nonuniform
    self.w_bits = list(w_bits) if w_bits is not None else None
    self.a_bits = list(a_bits) if a_bits is not None else None
    self.use_buckets, self.bucket_type, self.bucket_size = use_buckets, bucket_type, bucket_size
    self.codebooks: Dict[int, torch.Tensor] = {}        # matmul index -> codebook (NUQ)


class Scope(object):
  """Variables of one model scope + the TF-style execution helpers."""

  def __init__(self, values: Dict[str, np.ndarray], scope: str, trainable: bool = True):
    self.scope = scope
    self.v: Dict[str, torch.Tensor] = {}
    for name, val in values.items():
      if not name.startswith(scope + '/'):
        continue
      t = torch.from_numpy(np.array(val, dtype=np.float32, copy=True))
      is_state = name.endswith('moving_mean') or name.endswith('moving_variance')
      if trainable and not is_state:
        t.requires_grad_(True)
      self.v[name] = t
    self.training = True
    self.quant = QuantSpec()
    self.matmul_names: List[str] = []
    self.act_names: List[str] = []
    self.rec = None                  # graph trace of one forward (start_trace), the channel pruner's view of the network
    self.freeze_stats = False        # training-mode BN WITHOUT the moving-average update ops (sess.run of a feature tensor)
    self._begin()

  # -- graph trace: ops in creation order with TF's op types, tensors by '<op>:0' (oracle/cp_features_oracle.py) -------------
  def start_trace(self):
    self.rec = {'ops': [], 'ids': {}, 'tensors': {}, 'keep': [], 'convs': {}}

  def _r(self, type_, name, inputs, out):
    if self.rec is None:
      return out
    r, full = self.rec, self.scope + '/' + name
    r['ops'].append((full, type_, [r['ids'][id(t)] for t in inputs]))
    r['ids'][id(out)] = full
    r['tensors'][full + ':0'] = out
    r['keep'].append(out)            # keeps id() unique for the lifetime of the trace
    return out

  # -- bookkeeping: TF auto names, creation-order indices -----------------------------------------
  def _begin(self):
    self._names: Dict[str, int] = {}
    self.i_mm = 0
    self.i_act = 0
    self.matmul_names = []
    self.act_names = []

  def uname(self, base: str) -> str:
    k = self._names.get(base, 0)
    self._names[base] = k + 1
    return base if k == 0 else '%s_%d' % (base, k)

  def var(self, name: str) -> torch.Tensor:
    return self.v[self.scope + '/' + name]

  def trainable_names(self) -> List[str]:
    return [n for n, t in self.v.items() if t.requires_grad]

  # -- matmul ops ---------------------------------------------------------------------------------
  def _quant_weight(self, w: torch.Tensor, name: str) -> torch.Tensor:
    i = self.i_mm
    self.i_mm += 1
    self.matmul_names.append(self.scope + '/' + name)
    q = self.quant
    bits = q.w_bits[i] if (q.w_bits is not None and q.kind != 'none') else None
    if bits is None:
      return w
    if q.kind == 'uniform':
      return _WeightUQ.apply(w, int(bits), q.use_buckets, q.bucket_type, q.bucket_size)
    if q.kind == 'nonuniform':
      return _WeightNUQ.apply(w, q.codebooks[i], int(bits), q.use_buckets, q.bucket_type, q.bucket_size)
    raise ValueError(q.kind)

  @staticmethod
  def _same_pads(size, k, stride):
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return total // 2, total - total // 2

  def conv2d(self, x, name, stride=1, padding='SAME', bias_name=None, kernel_name='kernel'):
    """tf.layers.conv2d / slim.conv2d on NCHW-logical x with an HWIO kernel variable."""
    w = self._quant_weight(self.var(name + '/' + kernel_name), name + '/' + kernel_name)
    k = w.shape[0]
    x_in = x
    if padding == 'SAME':
      ph, pw = self._same_pads(x.shape[2], k, stride), self._same_pads(x.shape[3], k, stride)
      if any(ph) or any(pw):
        x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]))
    b = self.var(name + '/' + bias_name) if bias_name else None
    if self.rec is None:
      return F.conv2d(x, w.permute(3, 2, 0, 1), b, stride=stride)
    y = self._r('Conv2D', name + '/Conv2D', [x_in], F.conv2d(x, w.permute(3, 2, 0, 1), None, stride=stride))
    self.rec['convs'][self.scope + '/' + name + '/Conv2D'] = dict(
        name=self.scope + '/' + name + '/Conv2D', input=self.rec['ids'][id(x_in)] + ':0', h=int(w.shape[0]), w=int(w.shape[1]),
        c=int(w.shape[2]), strides=(stride, stride), padding=padding, kernel=self.scope + '/' + name + '/' + kernel_name)
    return y if b is None else self._r('BiasAdd', name + '/BiasAdd', [y], y + b.view(1, -1, 1, 1))

  def depthwise(self, x, name, stride=1, kernel_name='depthwise_weights'):
    w = self._quant_weight(self.var(name + '/' + kernel_name), name + '/' + kernel_name)   # [kh,kw,C,1]
    k, C = w.shape[0], w.shape[2]
    x_in = x
    ph, pw = self._same_pads(x.shape[2], k, stride), self._same_pads(x.shape[3], k, stride)
    if any(ph) or any(pw):
      x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]))
    return self._r('DepthwiseConv2dNative', name + '/depthwise', [x_in],
                   F.conv2d(x, w.permute(2, 3, 0, 1), None, stride=stride, groups=C))

  def dense(self, x, name):
    w = self._quant_weight(self.var(name + '/kernel'), name + '/kernel')        # [in, out]
    return self._r('MatMul', name + '/MatMul', [x], x @ w + self.var(name + '/bias'))

  # -- normalisation / activation -------------------------------------------------------------------
  def batch_norm(self, x, name, momentum, eps, names=('gamma', 'beta', 'moving_mean', 'moving_variance')):
    """tf.layers.batch_normalization(fused=True) over NCHW-logical x (resnet_model.py:55-62)."""
    gamma, beta = self.var(name + '/' + names[0]), self.var(name + '/' + names[1])
    mm, mv = self.var(name + '/' + names[2]), self.var(name + '/' + names[3])
    if self.training:
      n = x.numel() // x.shape[1]
      mean = x.mean(dim=(0, 2, 3))
      var = x.var(dim=(0, 2, 3), unbiased=False)
      if not self.freeze_stats:
        with torch.no_grad():                  # moving average fed the UNBIASED variance (fused BN)
          mm.mul_(momentum).add_(mean * (1 - momentum))
          mv.mul_(momentum).add_(var * (n / max(n - 1, 1)) * (1 - momentum))
    else:
      mean, var = mm, mv
    inv = torch.rsqrt(var + eps)
    scale = (gamma * inv).view(1, -1, 1, 1)
    shift = (beta - mean * gamma * inv).view(1, -1, 1, 1)
    return self._r('FusedBatchNorm', name + '/FusedBatchNorm', [x], x * scale + shift)

  def activation(self, u, kind, name):
    j = self.i_act
    self.i_act += 1
    self.act_names.append(self.scope + '/' + name)
    q = self.quant
    bits = q.a_bits[j] if (q.a_bits is not None and q.kind != 'none') else None
    if bits is None:
      return self._r(kind, name, [u], F.relu(u) if kind == 'Relu' else F.relu6(u))
    return _ActUQ.apply(u, int(bits), kind)


# =================================================================================================
# networks
# =================================================================================================

def lenet_forward(s: Scope, x, nb_classes):
  """nets/lenet_at_cifar10.py:34-68 (incl. the trailing softmax)."""
  x = s.conv2d(x, 'conv1', 1, 'VALID', bias_name='bias')
  a = s.activation(x, 'Relu', 'relu1')
  x = s._r('MaxPool', 'pool1/MaxPool', [a], F.max_pool2d(a, 2, 2))
  x = s.conv2d(x, 'conv2', 1, 'VALID', bias_name='bias')
  a = s.activation(x, 'Relu', 'relu2')
  x = s._r('MaxPool', 'pool2/MaxPool', [a], F.max_pool2d(a, 2, 2))
  x = s._r('Reshape', 'flatten/Reshape', [x], x.permute(0, 2, 3, 1).reshape(x.shape[0], -1))
  x = s.activation(s.dense(x, 'fc3'), 'Relu', 'relu3')
  x = s.dense(x, 'fc4')
  return torch.softmax(x, dim=1)


_RESNET_BN = (0.997, 1e-5)


def _conv_fixed_padding(s: Scope, x, k, strides):
  """conv2d_fixed_padding (resnet_model.py:92-103)."""
  name = s.uname('resnet_model/conv2d')
  if strides > 1:
    pt = k - 1
    pb = pt // 2
    if pt:
      x = s._r('Pad', name + '/Pad', [x], F.pad(x, (pb, pt - pb, pb, pt - pb)))
    elif s.rec is not None:          # fixed_padding() always emits its tf.pad, also with zero widths (kernel_size 1)
      x = s._r('Pad', name + '/Pad', [x], x.clone())
  return s.conv2d(x, name, strides, 'SAME' if strides == 1 else 'VALID')


def _bn_relu(s: Scope, x):
  name = s.uname('resnet_model/batch_normalization')
  x = s.batch_norm(x, name, *_RESNET_BN)
  return s.activation(x, 'Relu', s.uname('resnet_model/Relu'))


def resnet_v2_forward(s: Scope, x, cfg):
  """ResNet.Model.__call__ (resnet_model.py:489-554), resnet_version 2 only."""
  bottleneck = cfg['bottleneck']
  x = _conv_fixed_padding(s, x, cfg['kernel_size'], cfg['conv_stride'])
  if cfg['first_pool_size']:
    k, st = cfg['first_pool_size'], cfg['first_pool_stride']
    ph, pw = s._same_pads(x.shape[2], k, st), s._same_pads(x.shape[3], k, st)
    x_in = x
    if any(ph) or any(pw):
      x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]), value=float('-inf'))
    x = s._r('MaxPool', 'resnet_model/max_pooling2d/MaxPool', [x_in], F.max_pool2d(x, k, st))
  for i, nblocks in enumerate(cfg['block_sizes']):
    filters = cfg['num_filters'] * (2 ** i)
    for b in range(nblocks):
      strides = cfg['block_strides'][i] if b == 0 else 1
      shortcut = x
      y = _bn_relu(s, x)
      if b == 0:                             # projection shortcut is created FIRST (:295-298)
        shortcut = _conv_fixed_padding(s, y, 1, strides)
      if bottleneck:
        y = _conv_fixed_padding(s, y, 1, 1)
        y = _bn_relu(s, y)
        y = _conv_fixed_padding(s, y, 3, strides)
        y = _bn_relu(s, y)
        y = _conv_fixed_padding(s, y, 1, 1)
      else:
        y = _conv_fixed_padding(s, y, 3, strides)
        y = _bn_relu(s, y)
        y = _conv_fixed_padding(s, y, 3, 1)
      x = s._r('Add', s.uname('resnet_model/add'), [y, shortcut], y + shortcut)
  x = _bn_relu(s, x)
  x = s._r('Mean', 'resnet_model/Mean', [x], x.mean(dim=(2, 3)))
  return s.dense(x, 'resnet_model/dense')


_MBV1_DEFS = [('c', 3, 2, 32), ('d', 3, 1, 64), ('d', 3, 2, 128), ('d', 3, 1, 128), ('d', 3, 2, 256),
              ('d', 3, 1, 256), ('d', 3, 2, 512), ('d', 3, 1, 512), ('d', 3, 1, 512), ('d', 3, 1, 512),
              ('d', 3, 1, 512), ('d', 3, 1, 512), ('d', 3, 2, 1024), ('d', 3, 1, 1024)]
_MBV1_BN = (0.9997, 1e-3)
_SLIM_BN = ('gamma', 'beta', 'moving_mean', 'moving_variance')


def mobilenet_v1_forward(s: Scope, x, cfg):
  """mobilenet_v1 (utils/external/mobilenet_v1.py:168-392); dropout mask supplied by the caller."""
  sc = 'MobilenetV1'
  for i, (kind, k, stride, depth) in enumerate(_MBV1_DEFS):
    if kind == 'c':
      name = '%s/Conv2d_%d' % (sc, i)
      x = s.conv2d(x, name, stride, 'SAME', kernel_name='weights')
      x = s.batch_norm(x, name + '/BatchNorm', *_MBV1_BN, names=_SLIM_BN)
      x = s.activation(x, 'Relu6', name + '/Relu6')
    else:
      name = '%s/Conv2d_%d_depthwise' % (sc, i)
      x = s.depthwise(x, name, stride)
      x = s.batch_norm(x, name + '/BatchNorm', *_MBV1_BN, names=_SLIM_BN)
      x = s.activation(x, 'Relu6', name + '/Relu6')
      name = '%s/Conv2d_%d_pointwise' % (sc, i)
      x = s.conv2d(x, name, 1, 'SAME', kernel_name='weights')
      x = s.batch_norm(x, name + '/BatchNorm', *_MBV1_BN, names=_SLIM_BN)
      x = s.activation(x, 'Relu6', name + '/Relu6')
  kk = (min(x.shape[2], 7), min(x.shape[3], 7))
  x = s._r('AvgPool', sc + '/Logits/AvgPool_1a/AvgPool', [x], F.avg_pool2d(x, kk))
  mask = cfg.get('dropout_mask')
  if s.training and mask is not None:
    x = s._r('Mul', sc + '/Logits/Dropout_1b/dropout/mul', [x], x * torch.from_numpy(mask).view(x.shape[0], -1, 1, 1))
  x = s.conv2d(x, sc + '/Logits/Conv2d_1c_1x1', 1, 'SAME', bias_name='biases', kernel_name='weights')
  return x.reshape(x.shape[0], -1)


def resnet_cfg(dataset: str, resnet_size: int) -> Dict:
  """nets/resnet_at_cifar10.py:36-60 / nets/resnet_at_ilsvrc12.py:36-97."""
  if dataset == 'cifar_10':
    if resnet_size % 6 != 2:
      raise ValueError('resnet_size must be 6n + 2:', resnet_size)
    nb = (resnet_size - 2) // 6
    return dict(bottleneck=False, num_filters=16, kernel_size=3, conv_stride=1, first_pool_size=None,
                first_pool_stride=None, block_sizes=[nb] * 3, block_strides=[1, 2, 2])
  sizes = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3],
           200: [3, 24, 36, 3]}[resnet_size]
  return dict(bottleneck=resnet_size >= 50, num_filters=64, kernel_size=7, conv_stride=2, first_pool_size=3,
              first_pool_stride=2, block_sizes=sizes, block_strides=[1, 2, 2, 2])


# =================================================================================================
# the learner
# =================================================================================================

class OracleLearner(object):
  """One learner of the reference on the CPU.

  cfg keys (all mirror reference flags): model ('lenet'|'resnet'|'mobilenet_v1'), dataset
  ('cifar_10'|'ilsvrc_12'), resnet_size, nb_classes, learner ('full-prec'|'uniform'|'non-uniform'|
  'weight-sparse'|'channel'), loss_w_dcy, enbl_dst, loss_w_dst, tempr_dst, momentum, and the learner's own
  flags (uql_*, nuql_*, ws_*, cp_*).  'channel' = the masked fine-tune of the channel-pruned learner
  (cp learner.py:313-471): cfg['cp_fake_pruning'] = {kernel variable name: (keep_in[bool C_in], keep_out[bool C_out])}
  is the pruner's `fake_pruning_dict` re-keyed by variable, cfg['cp_optimizer'] = 'adam' (cp_finetune and not
  cp_retrain: AdamOptimizer(cp_lrn_rate_ft), :357-359) | 'momentum' (:360-362).  `lrn_rate(step)` is supplied by the caller (the schedules are
  checked separately against pf_oracle.*setup_bnds_decay_rates / lrn_rate_piecewise).
  """

  def __init__(self, values: Dict[str, np.ndarray], cfg: Dict, lrn_rate, teacher_values=None):
    self.cfg = dict(cfg)
    self.lrn_rate = lrn_rate
    self.student = Scope(values, 'model', trainable=True)
    self.teacher = None
    if cfg.get('enbl_dst'):
      tv = teacher_values
      if tv is None:                           # distillation_helper.py:122-145: renamed copy of ./models
        tv = {'distilled_model/' + '/'.join(k.split('/')[1:]): v for k, v in values.items()}
      self.teacher = Scope(tv, 'distilled_model', trainable=False)
      self.teacher.training = False
    self.step = 0
    self.kind = cfg.get('learner', 'full-prec')
    # dry run to learn the creation order of the ops
    self._dry_run()
    self.slots: Dict[str, List[np.ndarray]] = {}
    self.opt_kind = 'adam' if self.kind in ('uniform', 'non-uniform') else 'momentum'
    self.adam_t = 0
    self.masks: Dict[str, np.ndarray] = {}
    self.bkups: Dict[str, np.ndarray] = {}
    if self.kind == 'uniform':
      self._setup_uniform()
    elif self.kind == 'non-uniform':
      self._setup_nonuniform()
    elif self.kind == 'weight-sparse':
      self._setup_ws()
    elif self.kind == 'channel':
      self._setup_cp()
    self.opt_vars = self._select_opt_vars()

  # -- model dispatch ------------------------------------------------------------------------------
  def _forward(self, s: Scope, x_nhwc: torch.Tensor, training: bool, extra=None):
    s.training = training
    s._begin()
    x = x_nhwc.permute(0, 3, 1, 2)
    if s.rec is not None:
      s._r('Placeholder', 'mem_images', [], x)
    m = self.cfg['model']
    if m == 'lenet':
      return lenet_forward(s, x, self.cfg['nb_classes'])
    if m == 'resnet':
      return resnet_v2_forward(s, x, resnet_cfg(self.cfg['dataset'], self.cfg['resnet_size']))
    if m == 'mobilenet_v1':
      return mobilenet_v1_forward(s, x, extra or {})
    raise ValueError(m)

  def pruner_view(self, training: bool = True, extra=None):
    """The channel pruner's view of the STUDENT network (cp learner.py:250-255 builds it with forward_train, i.e. batch
    statistics; its sess.run of a feature tensor runs no moving-average update): (ops, convs, shapes, run) for
    oracle/cp_features_oracle.py -- `run(images NHWC, names) -> [NHWC float32 arrays]` re-executes the network."""
    s = self.student

    def trace(images):
      s.start_trace()
      s.freeze_stats = True
      try:
        with torch.no_grad():
          self._forward(s, torch.from_numpy(np.ascontiguousarray(images, dtype=np.float32)), training, extra)
        return s.rec
      finally:
        s.rec, s.freeze_stats = None, False

    def run(images, names):
      rec = trace(images)
      return [rec['tensors'][n].permute(0, 2, 3, 1).contiguous().numpy() if rec['tensors'][n].dim() == 4
              else rec['tensors'][n].numpy() for n in names]
    shape = self.cfg.get('image_shape', (32, 32, 3))
    rec = trace(np.zeros((2,) + tuple(shape), np.float32))
    shapes = {n: (int(t.shape[2]), int(t.shape[3]), int(t.shape[1])) for n, t in rec['tensors'].items() if t.dim() == 4}
    return rec['ops'], rec['convs'], shapes, run

  def _dry_run(self):
    shape = self.cfg.get('image_shape', (32, 32, 3))
    with torch.no_grad():
      self._forward(self.student, torch.zeros((2,) + tuple(shape)), False)
    self.n_matmul = len(self.student.matmul_names)
    self.n_act = len(self.student.act_names)
    self.matmul_var_names = list(self.student.matmul_names)

  # -- learner set-up ------------------------------------------------------------------------------
  def _quant_indices(self, all_layers: bool) -> List[int]:
    idx = list(range(self.n_matmul))
    return idx if all_layers else idx[1:-1]             # search_matmul_op (uq utils.py:115-125)

  def _setup_uniform(self):
    c = self.cfg
    sel = set(self._quant_indices(c.get('uql_quantize_all_layers', False)))
    wb = [c.get('uql_weight_bits', 4) if i in sel else None for i in range(self.n_matmul)]
    ab = [c.get('uql_activation_bits', 32)] * self.n_act    # always inserted (uq learner.py:336)
    self.student.quant = QuantSpec('uniform', wb, ab, c.get('uql_use_buckets', False),
                                   c.get('uql_bucket_type', 'channel'), c.get('uql_bucket_size', 256))

  def _setup_nonuniform(self):
    c = self.cfg
    sel = set(self._quant_indices(c.get('nuql_quantize_all_layers', False)))
    wb = [c.get('nuql_weight_bits', 4) if i in sel else None for i in range(self.n_matmul)]
    # activations: uniform fake-quant at nuql_activation_bits (nuq utils.py:58-85, 245-282)
    ab = [c.get('nuql_activation_bits', 32)] * self.n_act
    q = QuantSpec('nonuniform', wb, ab, c.get('nuql_use_buckets', False), c.get('nuql_bucket_type', 'split'),
                  c.get('nuql_bucket_size', 256))
    self.student.quant = q
    self.init_clusters()

  def init_clusters(self):
    """ops['cluster_init'] (nuq learner.py:128-129): codebooks from the CURRENT weights."""
    q, c = self.student.quant, self.cfg
    for i, name in enumerate(self.matmul_var_names):
      if q.w_bits[i] is None:
        continue
      w = self.student.v[name].detach().numpy()
      _, info = O.nuq_quantize(w, q.w_bits[i], None, q.use_buckets, q.bucket_type, q.bucket_size,
                               c.get('nuql_init_style', 'quantile'))
      cb = torch.from_numpy(np.ascontiguousarray(info['codebook']))
      cb.requires_grad_(True)
      q.codebooks[i] = cb

  def _setup_ws(self):
    c = self.cfg
    names = O.get_maskable_var_names(self.student.trainable_names())
    if c.get('ws_prune_ratio_prtl', 'uniform') == 'uniform':
      self.prune_ratios = O.pr_uniform(names, c['ws_prune_ratio'])
    else:
      nb = [self.student.v[n].numel() for n in names]
      self.prune_ratios = O.pr_heurist(names, nb, c['ws_prune_ratio'])
    for n in names:                                     # __build_masks (ws learner.py:277-280)
      self.masks[n] = np.ones(tuple(self.student.v[n].shape), dtype=np.float32)
      self.bkups[n] = self.student.v[n].detach().numpy().copy()

  def _setup_cp(self):
    """__calc_grads_pruned (cp learner.py:381-421): one constant {0,1} mask per pruned Conv2D kernel,
    ones(HWIO) with the pruned input rows and output columns zeroed; every other gradient passes unmasked
    (depthwise kernels are never in fake_pruning_dict).  The optimiser follows __build_pruned_train_model
    (:357-362); its slots are re-initialised by train_init_op (:376-377), i.e. start from zero as here."""
    c = self.cfg
    self.opt_kind = c.get('cp_optimizer', 'adam')
    for name, (keep_in, keep_out) in c.get('cp_fake_pruning', {}).items():
      self.masks[name] = O.cp_grad_mask(tuple(self.student.v[name].shape), keep_in, keep_out)

  def _select_opt_vars(self) -> List[str]:
    names = self.student.trainable_names()
    if self.kind == 'non-uniform':
      mode = self.cfg.get('nuql_opt_mode', 'weights')
      cbs = ['@codebook/%d' % i for i in sorted(self.student.quant.codebooks)]
      if mode == 'weights':
        return names
      if mode == 'cluster':
        return cbs
      if mode == 'both':
        return names + cbs
      raise ValueError(mode)
    return names

  def _tensor_of(self, name: str) -> torch.Tensor:
    if name.startswith('@codebook/'):
      return self.student.quant.codebooks[int(name.split('/')[1])]
    return self.student.v[name]

  # -- loss ---------------------------------------------------------------------------------------
  def _l2_names(self) -> List[str]:
    names = self.student.trainable_names()
    if self.cfg['model'] == 'resnet':
      return [n for n in names if 'batch_normalization' not in n]
    return names                                        # lenet :107, mobilenet :108-112: every trainable

  def _loss(self, labels: torch.Tensor, logits: torch.Tensor, logits_dst):
    c = self.cfg
    B = logits.shape[0]
    ce = -(labels * F.log_softmax(logits, dim=1)).sum(dim=1).sum() / B
    reg = sum(0.5 * (self.student.v[n] ** 2).sum() for n in self._l2_names())
    # "Strictly speaking, clusters should be not included for regularization" (nuq learner.py:219-220):
    # the `clusters` variables are trainable variables of the model scope, so they ARE in the L2 sum
    for cb in self.student.quant.codebooks.values():
      reg = reg + 0.5 * (cb ** 2).sum()
    model_loss = ce + c['loss_w_dcy'] * reg
    dst = None
    loss = model_loss
    if logits_dst is not None:
      T = c.get('tempr_dst', 4.0)
      soft = F.softmax(logits_dst / T, dim=1)
      dst = c.get('loss_w_dst', 4.0) * (-(soft * F.log_softmax(logits / T, dim=1)).sum(dim=1).sum() / B)
      loss = loss + dst
    return loss, model_loss, dst

  # -- one iteration --------------------------------------------------------------------------------
  def compute_grads(self, images: np.ndarray, labels: np.ndarray, extra=None):
    """Forward + backward of one rank's mini-batch: (losses dict, {variable: gradient}) -- BN moving statistics and
    activation ranges are this replica's own, as in the reference's data-parallel runs (no sync-BN)."""
    x = torch.from_numpy(np.asarray(images, dtype=np.float32))
    y = torch.from_numpy(np.asarray(labels, dtype=np.float32))
    logits_dst = None
    if self.teacher is not None:
      with torch.no_grad():
        logits_dst = self._forward(self.teacher, x, False)
    for n in self.opt_vars:
      self._tensor_of(n).grad = None
    logits = self._forward(self.student, x, True, extra)
    loss, model_loss, dst = self._loss(y, logits, logits_dst)
    loss.backward()
    grads = {}
    for n in self.opt_vars:
      t = self._tensor_of(n)
      grads[n] = t.grad.numpy().copy() if t.grad is not None else np.zeros(tuple(t.shape), np.float32)
    out = {'loss': float(loss.detach()), 'model_loss': float(model_loss.detach()),
           'dst_loss': None if dst is None else float(dst.detach()), 'logits': logits.detach().numpy()}
    return out, grads

  def apply_grads(self, grads) -> float:
    """apply_gradients of the (averaged) gradients: masks AFTER the reduction (ws learner.py:205-207), then the update."""
    lr = float(self.lrn_rate(self.step))
    if self.opt_kind == 'adam':
      self.adam_t += 1
    for n in self.opt_vars:
      t = self._tensor_of(n)
      g = grads[n]
      if n in self.masks:                               # __calc_grads_pruned (ws learner.py:314-332)
        g = O.masked_grad(g, self.masks[n])
      p = t.detach().numpy()
      if self.opt_kind == 'adam':
        m, v = self.slots.setdefault(n, [np.zeros_like(p), np.zeros_like(p)])
        p2, m2, v2 = O.adam_step(p, g, m, v, self.adam_t, lr)
        self.slots[n] = [m2, v2]
      else:
        (acc,) = self.slots.setdefault(n, [np.zeros_like(p)])
        p2, acc2 = O.momentum_step(p, g, acc, lr, self.cfg.get('momentum', 0.9))
        self.slots[n] = [acc2]
      with torch.no_grad():
        t.copy_(torch.from_numpy(np.ascontiguousarray(p2)))
    self.step += 1
    return lr

  def train_step(self, images: np.ndarray, labels: np.ndarray, extra=None) -> Dict:
    out, grads = self.compute_grads(images, labels, extra)
    out['lr'] = self.apply_grads(grads)
    return out

  def prune_step(self, nb_iters_train: int):
    """[prune_op, init_opt_op] (ws learner.py:124-131, 283-288): refresh masks, reset Momentum."""
    c = self.cfg
    for name, r_f in self.prune_ratios:
      r_t = O.ws_prune_ratio_dyn(self.step, nb_iters_train, r_f, c.get('ws_iter_ratio_beg', 0.1),
                                 c.get('ws_iter_ratio_end', 0.5), c.get('ws_prune_ratio_exp', 3.0))
      var = self.student.v[name].detach().numpy()
      var2, bk, mk, _ = O.ws_mask_refresh(var, self.bkups[name], self.masks[name], r_t)
      self.bkups[name], self.masks[name] = bk, mk
      with torch.no_grad():
        self.student.v[name].copy_(torch.from_numpy(np.ascontiguousarray(var2)))
    self.slots = {}

  def eval_batch(self, images: np.ndarray, labels: np.ndarray) -> Dict:
    x = torch.from_numpy(np.asarray(images, dtype=np.float32))
    y = torch.from_numpy(np.asarray(labels, dtype=np.float32))
    with torch.no_grad():
      logits = self._forward(self.student, x, False)
      logits_dst = self._forward(self.teacher, x, False) if self.teacher is not None else None
      loss, _, _ = self._loss(y, logits, logits_dst)
    out = logits.numpy()
    lab = y.numpy()
    metrics = O.metrics_ilsvrc(lab, out) if self.cfg['dataset'] == 'ilsvrc_12' else O.metrics_cifar(lab, out)
    return {'loss': float(loss), 'metrics': metrics, 'logits': out}

  def export(self) -> Dict[str, np.ndarray]:
    return {n: t.detach().numpy().copy() for n, t in self.student.v.items()}


# =================================================================================================
# CPU baseline for bench.py
# =================================================================================================

def _random_values_resnet(dataset, resnet_size, nb_classes, image_shape, seed=42):
  """Seeded variables for an oracle ResNet-v2 without touching the product package (same creation
  order and names as resnet_v2_forward)."""
  rng = np.random.RandomState(seed)
  values: Dict[str, np.ndarray] = {}
  cfg = resnet_cfg(dataset, resnet_size)

  def conv(name, k, cin, cout):
    std = np.sqrt(1.0 / (k * k * cin))
    values['model/' + name + '/kernel'] = (rng.randn(k, k, cin, cout) * std).astype(np.float32)

  def bn(name, c):
    values['model/' + name + '/gamma'] = np.ones(c, np.float32)
    values['model/' + name + '/beta'] = np.zeros(c, np.float32)
    values['model/' + name + '/moving_mean'] = np.zeros(c, np.float32)
    values['model/' + name + '/moving_variance'] = np.ones(c, np.float32)
  names: Dict[str, int] = {}

  def un(base):
    k = names.get(base, 0)
    names[base] = k + 1
    return base if k == 0 else '%s_%d' % (base, k)
  cin = cfg['num_filters']
  conv(un('resnet_model/conv2d'), cfg['kernel_size'], image_shape[2], cin)
  for i, nblocks in enumerate(cfg['block_sizes']):
    filters = cfg['num_filters'] * (2 ** i)
    cout = filters * 4 if cfg['bottleneck'] else filters
    for b in range(nblocks):
      bn(un('resnet_model/batch_normalization'), cin)
      if b == 0:
        conv(un('resnet_model/conv2d'), 1, cin, cout)
      if cfg['bottleneck']:
        conv(un('resnet_model/conv2d'), 1, cin, filters)
        bn(un('resnet_model/batch_normalization'), filters)
        conv(un('resnet_model/conv2d'), 3, filters, filters)
        bn(un('resnet_model/batch_normalization'), filters)
        conv(un('resnet_model/conv2d'), 1, filters, cout)
      else:
        conv(un('resnet_model/conv2d'), 3, cin, filters)
        bn(un('resnet_model/batch_normalization'), filters)
        conv(un('resnet_model/conv2d'), 3, filters, filters)
      cin = cout
  bn(un('resnet_model/batch_normalization'), cin)
  lim = np.sqrt(6.0 / (cin + nb_classes))
  values['model/resnet_model/dense/kernel'] = rng.uniform(-lim, lim, (cin, nb_classes)).astype(np.float32)
  values['model/resnet_model/dense/bias'] = np.zeros(nb_classes, np.float32)
  return values


def _random_values_mobilenet(nb_classes, depth_mult=1.0, min_depth=8, seed=42, in_channels=3):
  """Seeded variables of MobileNet-v1 in slim's naming (utils/external/mobilenet_v1.py)."""
  rng = np.random.RandomState(seed)
  vals: Dict[str, np.ndarray] = {}
  depth = lambda d: max(int(d * depth_mult), min_depth)

  def bn(prefix, c):
    vals[prefix + '/BatchNorm/gamma'] = (rng.rand(c) + 0.5).astype(np.float32)
    vals[prefix + '/BatchNorm/beta'] = (rng.randn(c) * 0.1).astype(np.float32)
    vals[prefix + '/BatchNorm/moving_mean'] = (rng.randn(c) * 0.1).astype(np.float32)
    vals[prefix + '/BatchNorm/moving_variance'] = (rng.rand(c) + 0.5).astype(np.float32)
  cin = in_channels
  for i, (kind, k, stride, d) in enumerate(_MBV1_DEFS):
    if kind == 'c':
      name = 'model/MobilenetV1/Conv2d_%d' % i
      vals[name + '/weights'] = (rng.randn(k, k, cin, depth(d)) * 0.2).astype(np.float32)
      bn(name, depth(d))
    else:
      name = 'model/MobilenetV1/Conv2d_%d_depthwise' % i
      vals[name + '/depthwise_weights'] = (rng.randn(k, k, cin, 1) * 0.3).astype(np.float32)
      bn(name, cin)
      name = 'model/MobilenetV1/Conv2d_%d_pointwise' % i
      vals[name + '/weights'] = (rng.randn(1, 1, cin, depth(d)) * (1.5 / np.sqrt(cin))).astype(np.float32)
      bn(name, depth(d))
    cin = depth(d)
  vals['model/MobilenetV1/Logits/Conv2d_1c_1x1/weights'] = (rng.randn(1, 1, cin, nb_classes) * 0.1).astype(np.float32)
  vals['model/MobilenetV1/Logits/Conv2d_1c_1x1/biases'] = (rng.randn(nb_classes) * 0.1).astype(np.float32)
  return vals


def net_fixture_recipe(model: str, dataset: str, resnet_size: int, nb_classes: int, image_shape, batch: int = 4,
                       seed: int = 7):
  """(values, images) of the network-definition fixtures (tests/golden/make_reference_golden.py executes the
  reference's own network code on them; tests/test_oracle_golden.py re-creates them with this recipe)."""
  rng = np.random.RandomState(seed + 1000)
  if model == 'resnet':
    vals = _random_values_resnet(dataset, resnet_size, nb_classes, image_shape, seed=seed)
  elif model == 'lenet':
    h, w, c = image_shape
    hh, ww = ((h - 4) // 2 - 4) // 2, ((w - 4) // 2 - 4) // 2
    vals = {'model/conv1/kernel': rng.randn(5, 5, c, 32) * 0.1, 'model/conv1/bias': rng.randn(32) * 0.1,
            'model/conv2/kernel': rng.randn(5, 5, 32, 64) * 0.05, 'model/conv2/bias': rng.randn(64) * 0.1,
            'model/fc3/kernel': rng.randn(hh * ww * 64, 256) * 0.03, 'model/fc3/bias': rng.randn(256) * 0.1,
            'model/fc4/kernel': rng.randn(256, nb_classes) * 0.1, 'model/fc4/bias': rng.randn(nb_classes) * 0.1}
    vals = {k: v.astype(np.float32) for k, v in vals.items()}
  elif model == 'mobilenet_v1':
    vals = _random_values_mobilenet(nb_classes, depth_mult=resnet_size / 100.0, seed=seed)
    images = rng.randn(batch, *image_shape).astype(np.float32)
    return vals, images
  else:
    raise ValueError(model)
  for k in sorted(vals):                       # non-trivial BN parameters / statistics / biases
    if 'moving_mean' in k:
      vals[k] = (rng.randn(*vals[k].shape) * 0.1).astype(np.float32)
    elif 'moving_variance' in k:
      vals[k] = (rng.rand(*vals[k].shape) + 0.5).astype(np.float32)
    elif k.endswith('gamma'):
      vals[k] = (rng.rand(*vals[k].shape) + 0.5).astype(np.float32)
    elif k.endswith('beta') or (k.endswith('bias') and model == 'resnet'):
      vals[k] = (rng.randn(*vals[k].shape) * 0.1).astype(np.float32)
  images = rng.randn(batch, *image_shape).astype(np.float32)
  return vals, images


def time_cpu_baseline(resnet_size=50, image_size=224, batch=4, steps=3, weight_bits=8, act_bits=8,
                      nb_classes=1001, enbl_dst=True, threads: Optional[int] = None, budget_s: float = 25.0) -> Dict:
  """Time the oracle UniformQuantLearner step (the reference's TF-CPU path restated; TF itself is
  unavailable) on the host cores.  Bounded sample: two warm-up steps (one if the budget is short), then up to `steps` timed steps while
  the wall-clock budget lasts; if the warm-up alone exhausts the budget it IS the sample.  Threads: the
  cores this process may run on (sched_getaffinity), capped at 32 -- an unconstrained 256-thread pool on a shared box is
  100x slower.  The record names the CPU model and the core count (the baseline is only comparable with them)."""
  import os
  try:
    avail = len(os.sched_getaffinity(0))
  except AttributeError:
    avail = os.cpu_count() or 1
  threads = threads or max(1, min(avail, 32))
  cpu_model = 'unknown'
  try:
    with open('/proc/cpuinfo') as f:
      for ln in f:
        if ln.startswith('model name'):
          cpu_model = ln.split(':', 1)[1].strip()
          break
  except OSError:
    pass
  torch.set_num_threads(threads)
  shape = (image_size, image_size, 3)
  values = _random_values_resnet('ilsvrc_12', resnet_size, nb_classes, shape)
  cfg = dict(model='resnet', dataset='ilsvrc_12', resnet_size=resnet_size, nb_classes=nb_classes, learner='uniform',
             uql_weight_bits=weight_bits, uql_activation_bits=act_bits, loss_w_dcy=1e-4, enbl_dst=enbl_dst,
             image_shape=shape)
  lrn = OracleLearner(values, cfg, lambda step: 1e-5)
  rng = np.random.RandomState(1234)
  means = np.array([123.68, 116.78, 103.94], dtype=np.float32)
  images = rng.randint(0, 256, size=(batch,) + shape).astype(np.float32) - means
  labels = np.zeros((batch, nb_classes), np.float32)
  labels[np.arange(batch), rng.randint(0, nb_classes, batch)] = 1
  t0 = time.perf_counter()
  lrn.train_step(images, labels)
  warm = time.perf_counter() - t0
  n_warm = 1
  if 3 * warm <= budget_s:                 # two warm-up steps where the budget allows a timed one behind them
    lrn.train_step(images, labels)
    n_warm, warm = 2, time.perf_counter() - t0
  done, dt = 0, 0.0
  t0 = time.perf_counter()
  while done < steps and warm + dt + (dt / done if done else warm / n_warm) <= budget_s:
    lrn.train_step(images, labels)
    done += 1
    dt = time.perf_counter() - t0
  if done == 0:
    done, dt, what = n_warm, warm, 'the %d warm-up step(s) only: they exhausted the %.0f s budget' % (n_warm, budget_s)
  else:
    what = '%d timed steps after %d warm-up' % (done, n_warm)
  return {'value': batch * done / dt, 'unit': 'images/s', 'cores': threads, 'cores_available': avail,
          'cpu_model': cpu_model, 'kind': 'port',
          'sample': 'oracle UniformQuantLearner step (NumPy fake-quant/loss/Adam + torch-CPU fp32 conv/BN), '
                    'ResNet-v2-%d %dx%d w%d/a%d%s, batch %d, %s; TF-1.x (the reference runtime) is not installable '
                    'here' % (resnet_size, image_size, image_size, weight_bits, act_bits,
                              ' + distillation' if enbl_dst else '', batch, what)}


def check_uq_step_against_oracle(learner, steps: int = 1, loss_rtol: float = 2e-4) -> float:
  """__graft_entry__.smoke(): run `steps` fine-tune steps of a freshly constructed (float32)
  UniformQuantLearner on the GPU and the same steps here, from the same variables and batches; returns
  the largest relative loss difference and raises if it exceeds `loss_rtol`."""
  from pocketflow_amd.flags import FLAGS          # the oracle may read the product's flags, never the reverse
  model = learner.model_name
  cfg = dict(model='lenet' if model == 'lenet' else 'resnet', dataset=learner.dataset_name,
             resnet_size=FLAGS.resnet_size if 'resnet_size' in FLAGS else 0, nb_classes=FLAGS.nb_classes,
             learner='uniform', loss_w_dcy=FLAGS.loss_w_dcy, enbl_dst=False, uql_weight_bits=FLAGS.uql_weight_bits,
             uql_activation_bits=FLAGS.uql_activation_bits, uql_use_buckets=FLAGS.uql_use_buckets,
             uql_bucket_type=FLAGS.uql_bucket_type, uql_bucket_size=FLAGS.uql_bucket_size,
             image_shape=tuple(learner.iter_train.batches[0][0].shape[1:]))
  ora = OracleLearner(learner.graph.store.export_numpy(), cfg, learner.lrn_rate)
  pool = [(i.cpu().numpy(), l.cpu().numpy()) for i, l in learner.iter_train.batches]
  worst = 0.0
  for step in range(steps):
    out = learner.train_step()
    ref = ora.train_step(*pool[step % len(pool)])
    rel = abs(float(out['loss'].detach()) - ref['loss']) / max(1.0, abs(ref['loss']))
    worst = max(worst, rel)
    if rel > loss_rtol:
      raise AssertionError('step %d: loss %.6f (HIP) vs %.6f (oracle)' % (step, float(out['loss'].detach()), ref['loss']))
  return worst
