"""Channel pruner (reference learners/channel_pruning/channel_pruner.py:52-808, He et al. 2017).

Host-side, rank 0 only, exactly as in the reference: feature maps are sampled at random spatial points
over `cp_nb_batches` batches, the input channels of each convolution are selected by a LASSO
(scikit-learn LassoLars with an alpha bisection, :456-577) and the kernel is re-fitted by least squares
(:443-454); pruning is "fake" -- tensors keep their shape, pruned channels are zeroed and recorded in
`fake_pruning_dict[op] = [keep_in, keep_out]` (:665-725) for the masked fine-tune.

What runs on the MI355X: the `cp_nb_batches` x (1 + #layers) forward passes that produce the sampled
feature maps and convolution inputs (the executor's tap mode, graph.Graph.taps); the LASSO / least
squares stay in scikit-learn on the host like in the reference.

Topologies: chains of Conv2D / DepthwiseConv2dNative with BN / ReLU(6) / pooling between them (MobileNet-v1 =
BASELINE config 3, LeNet-like nets) and residual networks: a convolution fed by a residual sum is not
"W1-prunable" (its producer is not a single convolution, :343-370) and a convolution whose output enters a
residual sum is re-fitted against `Y + residual_branch_diff` (:579-586, 611-614).

RL mode (`cp_prune_option auto`, :118-213): one 8-number state per convolution
[layer, n, c, H, W, stride, maxreduce, layercomp] / column maximum, the action (preserve ratio) constrained so
that the FLOP target `cp_preserve_ratio` stays reachable when every later layer is pruned to the lower bound.

Sampling modes (`--cp_sampling`).  'reference' reproduces channel_pruner.py:215-227, 263-341 to the letter: one pair of
point vectors per TENSOR NAME in the order [conv output, the Add it closes, ...] over the convolutions in creation
order, i.e. the residual sums get their OWN points (the correction `Y + residual_branch_diff` of :611-614 then adds
values taken at other pixels than Y's), and the projection shortcut of a block counts as "last in the block" like its
conv3 (its only consumer is the Add, model_wrapper.py:304-341).  'aligned' (default) samples a residual sum at the
points of the convolution it corrects and applies the correction to the block's last convolution only.
Feature BN mode (`--cp_feature_bn`): 'train' = batch statistics without moving-average updates, what the reference's
pruner graph computes (it is built with forward_train, learner.py:250-255, and sess.run of a feature tensor runs no
update op); 'inference' (default) = moving statistics.  With both switches on the reference setting the sampled
features, convolution inputs and residual diffs equal oracle/cp_features_oracle.py (tests/test_learners_cpu.py,
tests/test_learner_gpu.py).  The sampling generator is seeded (`cp_seed`; the reference uses the global np.random).
"""
from __future__ import annotations

import logging
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.graph import Conv2D, DepthwiseConv2D, Graph, to_device_images

flags.DEFINE_boolean('cp_lasso', True, 'If True use lasso and reconstruction otherwise prune according to weight magnitude')
flags.DEFINE_boolean('cp_quadruple', False, 'Restric the channels after pruning is a mutiple of 4')
flags.DEFINE_string('cp_reward_policy', 'accuracy', "'accuracy': best accuracy under the FLOP target | 'flops': fewest FLOPs at guaranteed accuracy")
flags.DEFINE_integer('cp_nb_points_per_layer', 10, 'Sample how many point for each layer')
flags.DEFINE_integer('cp_nb_batches', 30, 'Input how many bathes data into a model')
flags.DEFINE_integer('cp_seed', 2018, 'seed of the host-side sampling (the reference never seeds np.random)')
flags.DEFINE_string('cp_sampling', 'aligned', "'aligned': residual sums sampled at their convolution's points | 'reference': channel_pruner.py:263-341 to the letter")
flags.DEFINE_string('cp_feature_bn', 'inference', "BN statistics of the pruner's forward passes: 'inference' (moving) | 'train' (batch statistics, no update: the reference)")

log = logging.getLogger('pocketflow_amd')


def compute_pruned_kernel(X, W2, Y, c_new, rng, alpha=1e-4, tolerance=0.02, quadruple=False):
  """LASSO channel selection + least-squares reconstruction (reference :456-577).

  X [n, kh, kw, cin] sampled input patches, W2 [kh, kw, cin, cout] (HWIO), Y [n, cout] target outputs.
  Returns (idxs bool[cin], newW2 [cout, kh*kw*c_kept] = LinearRegression.coef_)."""
  from sklearn.linear_model import LassoLars, LinearRegression
  nb_samples, c_in, c_out = X.shape[0], X.shape[-1], W2.shape[-1]
  samples = rng.randint(0, nb_samples, min(400, nb_samples // 20))
  reshape_X = np.rollaxis(np.transpose(X, (0, 3, 1, 2)).reshape((nb_samples, c_in, -1))[samples], 1, 0)
  reshape_W2 = np.transpose(np.transpose(W2, (3, 2, 0, 1)).reshape((c_out, c_in, -1)), [1, 2, 0])
  product = np.matmul(reshape_X, reshape_W2).reshape((c_in, -1)).T
  reshape_Y = Y[samples].reshape(-1)
  solver = LassoLars(alpha=alpha, fit_intercept=False, max_iter=3000)

  def solve(a):
    solver.alpha = a
    solver.fit(product, reshape_Y)
    idx = solver.coef_ != 0.
    return idx, int(sum(idx)), solver.coef_

  if c_new == c_in:
    idxs = np.array([True] * c_new)
  else:
    left, right = 0, alpha
    lbound = c_new - tolerance * c_in / 2
    rbound = c_new + tolerance * c_in / 2
    while True:
      _, tmp, _ = solve(right)
      if tmp < c_new:
        break
      right *= 2
    while True:
      if lbound < 0:
        lbound = 1
      idxs, tmp, _ = solve(alpha)
      if quadruple and tmp % 4 == 0 and abs(tmp - lbound) <= 2:
        break
      if lbound <= tmp <= rbound:
        if quadruple:
          if tmp % 4 == 0:
            break
          elif tmp % 4 <= 2:
            rbound = tmp - 1
            lbound = lbound - 2
          else:
            lbound = tmp + 1
            rbound = rbound + 2
        else:
          break
      elif abs(left - right) <= right * 0.1:
        if lbound > 1:
          lbound = lbound - 1
        if rbound < c_in:
          rbound = rbound + 1
        left = left / 1.2
        right = right * 1.2
      elif tmp > rbound:
        left = left + (alpha - left) / 2
      else:
        right = right - (right - alpha) / 2
      if alpha < 1e-10:
        break
      alpha = (left + right) / 2
  if not np.any(idxs):
    # degenerate LASSO (every coefficient shrunk to zero: tiny sample or dead layer).  The reference would
    # crash in LinearRegression; keep the c_new channels with the largest kernel magnitude instead.
    order = np.argsort(-np.abs(W2).sum((0, 1, 3)))
    idxs = np.zeros(c_in, bool)
    idxs[order[:max(int(c_new), 1)]] = True
  reg = LinearRegression(n_jobs=-1, copy_X=True, fit_intercept=False)
  reg.fit(X[:, :, :, idxs].reshape((nb_samples, -1)), Y)
  return idxs, reg.coef_


class ChannelPruner(object):  # pylint: disable=too-many-instance-attributes
  """Prunes the convolutions of one Graph in creation order; `compress(ratio)` handles one layer."""

  FEATURE_NAMES = ['layer', 'n', 'c', 'H', 'W', 'stride', 'maxreduce', 'layercomp']

  def __init__(self, graph: Graph, forward_eval, batches, sm_writer=None, lbound=0, calc_loss=None,
               trainable_vars=None, forward_train=None):
    self.graph = graph
    self.forward_eval = forward_eval
    self.forward_train = forward_train
    if FLAGS.cp_sampling not in ('aligned', 'reference') or FLAGS.cp_feature_bn not in ('inference', 'train'):
      raise ValueError('cp_sampling: aligned | reference, cp_feature_bn: inference | train')
    if FLAGS.cp_feature_bn == 'train' and forward_train is None:
      raise ValueError("cp_feature_bn 'train' needs the learner's forward_train")
    self.batches = batches                       # list of (images NHWC float32 device tensor, labels)
    self.sm_writer = sm_writer
    self.lbound = lbound
    self.calc_loss, self.trainable_vars = calc_loss, trainable_vars
    self.rng = np.random.RandomState(FLAGS.cp_seed)
    self.state = 0
    self.points: Dict = {}
    self.feats_dict: Dict[object, np.ndarray] = {}
    self.feats_add: Dict[object, np.ndarray] = {}
    self.max_reduced_flops = 0
    self.preserve_ratio = None
    self.__trace()
    self.initialize_state()

  # -- topology ------------------------------------------------------------------------------------
  def __run(self, images, features=False):
    """One forward in tap mode; returns {layer: (input, output, producer, residual sum, residual operand)}.  `features`
    and cp_feature_bn 'train': batch statistics as in the reference's pruner graph (forward_train, learner.py:250-255) --
    the training path of the BN layers runs (it needs grad mode), the moving statistics it updates are put back."""
    g = self.graph
    g.taps = OrderedDict()
    try:
      if features and FLAGS.cp_feature_bn == 'train':
        state = g.store.state.clone()
        # a training forward has two more side effects than the BN moving statistics: a network with dropout (MobileNet) advances its
        # (seed, step)-keyed mask counter -- the later fine-tune must draw the masks it would draw without this pass
        steps = {k: n.dropout_step for k, n in getattr(g, 'nets', {}).items() if hasattr(n, 'dropout_step')}
        try:
          with torch.enable_grad(), g.as_default():
            self.logits = self.forward_train(to_device_images(images, g)).detach()
        finally:
          g.store.state.copy_(state)
          for k, v in steps.items():
            g.nets[k].dropout_step = v
        return OrderedDict((l, tuple(t.detach() if torch.is_tensor(t) else t for t in v)) for l, v in g.taps.items())
      with torch.no_grad(), g.as_default():
        self.logits = self.forward_eval(to_device_images(images, g))
      return g.taps
    finally:
      g.taps = None

  def __trace(self):
    taps = self.__run(self.batches[0][0])
    order = {id(op): i for i, op in enumerate(self.graph.matmul_ops)}
    # creation order = the order of g.get_operations() in the reference (the projection shortcut of a block is created
    # BEFORE its conv1, resnet_model.py:295-298, whatever order the executor calls them in)
    self.layers = sorted(taps.keys(), key=lambda l: order[id(l.op)])
    self.thisconvs: List[Conv2D] = [l for l in self.layers if isinstance(l, Conv2D)]
    self.fathers: Dict[object, Optional[object]] = {l: taps[l][2] for l in self.layers}
    self.out_hw = {l: (taps[l][1].shape[2], taps[l][1].shape[3]) for l in self.layers}
    self.last_in_resblock = {l for l in self.thisconvs if taps[l][3] is not None}
    # every convolution whose output enters a residual sum DIRECTLY -> the convolution the executor forms that sum in:
    # the block's last convolution itself and the projection shortcut (model_wrapper.py:304-341 names both)
    self.add_owner: Dict[object, object] = {}
    for l in self.thisconvs:
      if taps[l][3] is not None:
        self.add_owner[l] = l
        for p in self.thisconvs:
          if taps[p][1] is taps[l][4]:
            self.add_owner[p] = l
    self.names = [c.op.name for c in self.thisconvs]

  def is_W1_prunable(self, conv) -> bool:
    """The input reaches `conv` from another convolution through BN / ReLU / pooling only (:343-370)."""
    return self.fathers.get(conv) is not None

  def finallayer(self, offset=1):
    return len(self.thisconvs) - offset == self.state

  # -- RL state (:118-164) ----------------------------------------------------------------------------
  def getState(self, conv):
    kh, kw, c, n = conv.kernel.ref_shape
    H, W = self.out_hw[conv]
    return [self.state, n, c, H, W, conv.stride, 1., self.compute_layer_flops(conv)]

  def initialize_state(self):
    """States of all layers, FLOP targets, and the optimistic strategy table `max_strategy_dict[op] =
    [input preserve ratio, output preserve ratio]` (lbound where a later decision may still shrink it)."""
    self.best = -np.inf
    self.bestinfo = []
    allstate = []
    for self.state, conv in enumerate(self.thisconvs):
      allstate.append(self.getState(conv))
    self.state = 0
    states = np.array(allstate, dtype=np.float64)
    self.states = states / states.max(axis=0)
    self.layer_flops = self.states[:, 7].copy()
    self.model_flops = self.compute_model_flops()
    log.info('The original model flops is {}'.format(self.model_flops))
    self.currentStates = self.states.copy()
    self.desired_reduce = (1 - FLAGS.cp_preserve_ratio) * self.model_flops
    self.desired_preserve = FLAGS.cp_preserve_ratio * self.model_flops
    self.max_strategy_dict = {}
    self.fake_pruning_dict = {}
    for i, conv in enumerate(self.thisconvs):
      if self.is_W1_prunable(conv):
        father = self.fathers[conv]
        if isinstance(father, DepthwiseConv2D):
          if self.is_W1_prunable(father) and self.fathers[father].op.name in self.max_strategy_dict:
            self.max_strategy_dict[self.fathers[father].op.name][1] = self.lbound
        else:
          self.max_strategy_dict[father.op.name][1] = self.lbound
      inner = not (i == 0 or i == len(self.thisconvs) - 1)
      self.max_strategy_dict[conv.op.name] = [self.lbound, 1.] if inner else [1., 1.]
      cin, cout = conv.kernel.ref_shape[2], conv.kernel.ref_shape[3]
      self.fake_pruning_dict[conv.op.name] = [[True] * cin, [True] * cout]

  def __action_constraint(self, action):
    """Clip the preserve ratio so that the FLOP target stays reachable (:166-213)."""
    action = min(max(float(np.asarray(action).reshape(-1)[0]), 0.), 1.)
    if self.finallayer():
      return 1
    conv_op = self.thisconvs[self.state]
    prunable = self.is_W1_prunable(conv_op)
    father = self.fathers[conv_op] if prunable else None
    this_flops, other_flops = 0, 0
    for conv in self.thisconvs:
      curr_flops = self.compute_layer_flops(conv)
      strategy = self.max_strategy_dict[conv.op.name]
      if prunable and conv is father:
        this_flops += curr_flops * strategy[0]
      elif conv is conv_op:
        this_flops += curr_flops * strategy[1]
      else:
        other_flops += curr_flops * strategy[0] * strategy[1]
    self.max_reduced_flops = other_flops + this_flops * action
    if FLAGS.cp_reward_policy != 'accuracy' or self.state == 0:
      return action
    recommand_action = (self.desired_preserve - other_flops) / this_flops
    log.info('max_reduced_flops {} | desired_preserve {} | this flops {} | recommand action {}'.format(
        self.max_reduced_flops, self.desired_preserve, this_flops, recommand_action))
    return np.minimum(action, recommand_action)

  # -- flops -----------------------------------------------------------------------------------------
  def compute_layer_flops(self, conv) -> float:
    kh, kw, cin, cout = conv.kernel.ref_shape
    ho, wo = self.out_hw[conv]
    return 2.0 * ho * wo * kh * kw * cin * cout

  def compute_model_flops(self, fake=False) -> float:
    """FLOPs of the Conv2D layers (depthwise layers are not counted, as in the reference: :240-254 iterates
    `get_operations_by_type()` = Conv2D); `fake`: scaled by the current strategy table."""
    flops = 0.0
    for conv in self.thisconvs:
      f = self.compute_layer_flops(conv)
      if fake:
        f *= self.max_strategy_dict[conv.op.name][0] * self.max_strategy_dict[conv.op.name][1]
      flops += f
    return flops

  # -- sampling ----------------------------------------------------------------------------------------
  @staticmethod
  def __sample(y, xs, ys):
    sel = y[:, :, torch.as_tensor(xs, device=y.device), torch.as_tensor(ys, device=y.device)]
    return sel.permute(0, 2, 1).reshape(-1, y.shape[1]).float().cpu().numpy().astype(np.float64)

  def extract_features(self):
    """Outputs of every convolution (and of the residual sum it feeds, if any) of the ORIGINAL model at
    cp_nb_points_per_layer random points per batch (the same points for every image of a batch), over
    cp_nb_batches batches (:263-341).  cp_sampling 'reference': one draw per tensor NAME in the reference's order."""
    npts = FLAGS.cp_nb_points_per_layer
    nb_batches = min(FLAGS.cp_nb_batches, len(self.batches))
    ref = FLAGS.cp_sampling == 'reference'
    feats = {c: [] for c in self.thisconvs}
    adds = {c: [] for c in self.last_in_resblock}
    self.points = {}
    for b in range(nb_batches):
      taps = self.__run(self.batches[b][0], features=True)
      for conv in self.thisconvs:
        h, w = self.out_hw[conv]
        xs = self.rng.randint(0, h, npts)
        ys = self.rng.randint(0, w, npts)
        self.points[(b, conv)] = (xs.copy(), ys.copy())
        feats[conv].append(self.__sample(taps[conv][1], xs, ys))       # logical NCHW
        owner = self.add_owner.get(conv) if ref else (conv if conv in adds else None)
        if owner is None or (b, 'add', owner) in self.points:          # names are de-duplicated, first position kept (:299)
          continue
        if ref:                                                        # the Add tensor is a name of its own: own points
          xs = self.rng.randint(0, h, npts)
          ys = self.rng.randint(0, w, npts)
        self.points[(b, 'add', owner)] = (xs.copy(), ys.copy())
        if owner in taps and taps[owner][3] is not None:
          adds[owner].append(self.__sample(taps[owner][3], xs, ys))
    self.feats_dict = {c: np.vstack(v) for c, v in feats.items()}
    self.feats_add = {c: np.vstack(v) for c, v in adds.items()}
    self.nb_batches = nb_batches

  def residual_branch_diff(self, conv):
    """Change of the residual sum behind `conv` caused by the pruning done so far (:579-586)."""
    log.info("approximating residual branch diff")
    owner = self.add_owner[conv]
    cur = []
    for b in range(self.nb_batches):
      xs, ys = self.points[(b, 'add', owner)]
      cur.append(self.__sample(self.__run(self.batches[b][0], features=True)[owner][3], xs, ys))
    return self.feats_add[owner] - np.vstack(cur)

  def accuracy(self):
    """Mean `accuracy` metric of the current (partially pruned) model over the cached batches (:414-434)."""
    acc_list, rows, names = [], [], []
    for b in range(self.nb_batches if self.points else min(FLAGS.cp_nb_batches, len(self.batches))):
      images, labels = self.batches[b]
      self.__run(images)
      with torch.no_grad(), self.graph.as_default():
        __, metrics = self.calc_loss(labels.to(self.logits.device), self.logits, self.trainable_vars)
      acc_list.append(float(metrics['accuracy']))
      names = list(metrics.keys())
      rows.append([float(v) for v in metrics.values()])
    for k, v in zip(names, np.mean(np.array(rows), axis=0)):
      log.info('{}: {}'.format(k, v))
    return float(np.mean(acc_list))

  def __extract_input(self, conv) -> np.ndarray:
    """Input patches [n, kh, kw, cin] of `conv` at its sampled output points, from the CURRENT
    (partially pruned) model (:391-412)."""
    kh, kw, cin, _ = conv.kernel.ref_shape
    Xs = []
    for b in range(self.nb_batches):
      x = self.__run(self.batches[b][0], features=True)[conv][0]          # logical NCHW, materialised
      xs, ys = self.points[(b, conv)]
      if kh == 1 and kw == 1:
        ix = torch.as_tensor(xs * conv.stride, device=x.device)
        iy = torch.as_tensor(ys * conv.stride, device=x.device)
        p = x[:, :, ix, iy].permute(0, 2, 1).reshape(-1, 1, 1, cin)
      else:
        from pocketflow_amd.graph import _same_pads
        if conv.padding == 'SAME':
          ph, pw = _same_pads(x.shape[2], kh, conv.stride), _same_pads(x.shape[3], kw, conv.stride)
        elif isinstance(conv.padding, int):
          ph = pw = (conv.padding, conv.padding)
        else:
          ph = pw = (0, 0)
        xp = torch.nn.functional.pad(x, (pw[0], pw[1], ph[0], ph[1]))
        pts = []
        for px, py in zip(xs, ys):
          h0, w0 = int(px) * conv.stride, int(py) * conv.stride
          pts.append(xp[:, :, h0:h0 + kh, w0:w0 + kw].permute(0, 2, 3, 1))     # [B, kh, kw, cin]
        p = torch.stack(pts, dim=1).reshape(-1, kh, kw, cin)
      Xs.append(p.float().cpu().numpy().astype(np.float64))
    return np.vstack(Xs)

  # -- one layer ------------------------------------------------------------------------------------------
  def prune_kernel(self, conv, ratio):
    """Select the input channels of `conv` to keep and re-fit its kernel (:588-640)."""
    kh, kw, c, _ = conv.kernel.ref_shape
    nb_channel_new = max(int(np.around(c * ratio)), 1)
    newX = self.__extract_input(conv)
    Y = self.feats_dict[conv]
    # the output feeds a residual sum (:611-614); 'reference' sampling: also for the projection shortcut of the block
    if conv in self.feats_add or (FLAGS.cp_sampling == 'reference' and conv in self.add_owner):
      Y = Y + self.residual_branch_diff(conv)
    W2 = conv.kernel.to_ref(conv.kernel.master.detach().float().cpu().numpy()).astype(np.float64)
    if FLAGS.cp_lasso:
      idxs, newW2 = compute_pruned_kernel(newX, W2, Y, nb_channel_new, self.rng, quadruple=FLAGS.cp_quadruple)
    else:
      from sklearn.linear_model import LinearRegression
      order = np.argsort(-np.abs(W2).sum((0, 1, 3)))
      idxs = np.zeros(len(order), bool)
      idxs[order[:nb_channel_new]] = True
      reg = LinearRegression(fit_intercept=False)
      reg.fit(newX[:, :, :, idxs].reshape(newX.shape[0], -1), Y)
      newW2 = reg.coef_
    rel = lambda A, B: np.mean((A - B) ** 2) ** .5 / max(np.mean(A ** 2) ** .5, 1e-30)
    log.info('Prune {} c_in from {} to {} | feature map rmse {:.4f}'.format(
        conv.op.name, newX.shape[-1], int(sum(idxs)),
        rel(newX[:, :, :, idxs].reshape(newX.shape[0], -1).dot(newW2.T), Y)))
    kept = int(sum(idxs))
    newW2 = np.transpose(newW2.reshape(-1, kh, kw, kept), (1, 2, 3, 0))      # [kh, kw, kept, cout]
    return idxs, newW2, kept / float(len(idxs))

  def __assign(self, var, ref_value):
    var.master.copy_(torch.from_numpy(var.to_storage(ref_value.astype(np.float32))).to(var.master.device))

  def prune_W1(self, father, idxs):
    """Zero the pruned OUTPUT channels (and bias entries) of the producing convolution (:665-697)."""
    name = father.op.name
    if name not in self.fake_pruning_dict:              # a depthwise producer that is not W1-prunable itself
      c = father.kernel.ref_shape[2]
      self.fake_pruning_dict[name] = [[True] * c, [True] * c]
      self.max_strategy_dict[name] = [1.0, 1.0]
    self.max_strategy_dict[name][1] = sum(idxs) / float(len(idxs))
    self.fake_pruning_dict[name][1] = list(idxs)
    w = father.kernel.to_ref(father.kernel.master.detach().cpu().numpy())
    not_idxs = np.invert(np.asarray(idxs, dtype=bool))
    if isinstance(father, DepthwiseConv2D):
      w[:, :, not_idxs, :] = 0
    else:
      w[:, :, :, not_idxs] = 0
    self.__assign(father.kernel, w)
    bias = getattr(father, 'bias', None)
    if bias is not None:
      b = bias.master.detach().cpu().numpy().copy()
      b[not_idxs] = 0
      bias.master.copy_(torch.from_numpy(b).to(bias.master.device))

  def prune_W2(self, conv, idxs, W2=None):
    """Write the reconstructed kernel, zero the pruned INPUT channels (:699-725)."""
    name = conv.op.name
    self.max_strategy_dict[name][0] = sum(idxs) / float(len(idxs))
    self.fake_pruning_dict[name][0] = list(idxs)
    w = conv.kernel.to_ref(conv.kernel.master.detach().cpu().numpy())
    idxs = np.asarray(idxs, dtype=bool)
    if W2 is not None:
      w[:, :, idxs, :] = W2
    w[:, :, np.invert(idxs), :] = 0
    self.__assign(conv.kernel, w)

  def compress(self, c_ratio):
    """Prune the layer at `self.state` with preserve ratio `c_ratio` (:727-799).
    Returns (state of the next layer, [accuracy, pruned flops] (final layer) or [0, 1], done, applied ratio)."""
    auto = FLAGS.cp_prune_option == 'auto'
    if self.state == 0:
      c_ratio = 1.0                                        # first layer is not prunable
      if self.calc_loss is not None:
        self.accuracy()
    if self.finallayer():
      c_ratio = 1                                          # final layer is not prunable
    if auto:
      log.info('preserve ratio before constraint {}'.format(c_ratio))
      c_ratio = self.__action_constraint(c_ratio)
      log.info('preserve ratio after constraint {}'.format(c_ratio))
    conv = self.thisconvs[self.state]
    if c_ratio == 1:
      if auto:
        self.max_strategy_dict[conv.op.name][0] = c_ratio
        if self.is_W1_prunable(conv) and self.fathers[conv].op.name in self.max_strategy_dict:
          self.max_strategy_dict[self.fathers[conv].op.name][1] = c_ratio
    else:
      idxs, W2, c_ratio = self.prune_kernel(conv, c_ratio)
      if self.is_W1_prunable(conv):
        father = self.fathers[conv]
        while isinstance(father, DepthwiseConv2D):
          if self.is_W1_prunable(father):
            father = self.fathers[father]
          else:
            break
        log.info('father conv {}'.format(father.op.name))
        self.prune_W1(father, idxs)
      self.prune_W2(conv, idxs, W2)
      self.graph.store.sync_compute()
    log.info('Channel pruning the {} layer, the pruning rate is {}'.format(conv.op.name, c_ratio))

    if self.finallayer():
      acc = self.accuracy() if self.calc_loss is not None else 0.0
      pruned_flops = self.compute_model_flops(fake=True)
      self.preserve_ratio = pruned_flops / self.model_flops
      log.info('The accuracy is {} and the flops after pruning is {}'.format(acc, pruned_flops))
      log.info('The speedup ratio is {} | the original model flops is {}'.format(self.preserve_ratio, self.model_flops))
      log.info('The max strategy dict is {}'.format(self.max_strategy_dict))
      return self.currentStates[self.state].copy(), [acc, pruned_flops], True, c_ratio

    self.state += 1
    if auto:
      self.currentStates[self.state, 6] = self.max_reduced_flops / self.model_flops        # 'maxreduce'
    return self.currentStates[self.state].copy(), [0, 1], False, c_ratio
