"""Channel pruning with on-device channel selection (reference learners/channel_pruning_gpu/learner.py:110-568;
SURVEY 8f rank 4 / Appendix D.2).

Two copies of the network live on the device: `model` (full, constant) and `pruned_model`.  For every convolution i
(head & tail skipped by default) the input channels are selected by stochastic proximal gradient descent on the
layer's regression loss  L_i = l2_loss(conv_i(full) - conv_i(pruned)):

    W' = W - eta * dL_i/dW;   n_c = || W'[:, :, c, :] ||_2;   tau = percentile(n, p_t);   W <- W' * max(1 - tau / n_c, 0)

with p_t ramping linearly to the layer's target ratio over cpg_nb_iters_layer iterations and eta multiplied by 1.4 / 0.7
according to whether L_i fell (:445-493); the channels whose norm reached zero are frozen (mask) and the layer is
re-fitted with Adam(cpg_lrn_rate_adam) under `grad * mask` (:494-503); finally the whole network is fine-tuned with
Momentum and masked gradients (:153-176, 404-443).

Device mapping: the two partial forwards of a step run in tap mode up to layer i (learners/layerwise.py), the gradient
is ONE convolution backward, the group norms are a strided reduction over the KRSC kernel (axis I of [O][RS][I]), the
masks live in one flat buffer parallel to the kernel buffer and are applied inside the fused optimiser kernels
(pf_adam_flat / pf_momentum_flat), the whole-network fine-tune is the regular hot path.

Deviations: the full network's BN moving statistics are restored after every step (the reference never runs its
update ops); BN update ops of the pruned network behind layer i are not run during layer i's steps (the reference
runs all of them, which needs the whole forward pass); `cpg_seed`-less: data order comes from the seeded iterator.
"""
from __future__ import annotations

import logging
import os
from timeit import default_timer as timer

import numpy as np
import torch
import torch.distributed as dist

from pocketflow_amd import hip
from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.graph import Conv2D
from pocketflow_amd.learners.abstract_learner import AbstractLearner
from pocketflow_amd.learners.distillation_helper import DistillationHelper
from pocketflow_amd.learners.layerwise import forward_tapped, layers_of_vars
from pocketflow_amd.learners.weight_sparsification.learner import calc_prune_ratio
from pocketflow_amd.optim import FlatOptimizer
from pocketflow_amd.utils import checkpoint
from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw

flags.DEFINE_string('cpg_save_path', './models_cpg/model.ckpt', 'CPG: model\'s save path')
flags.DEFINE_string('cpg_save_path_eval', './models_cpg_eval/model.ckpt', 'CPG: model\'s save path for evaluation')
flags.DEFINE_string('cpg_prune_ratio_type', 'uniform', 'CPG: pruning ratio type (\'uniform\' OR \'list\')')
flags.DEFINE_float('cpg_prune_ratio', 0.5, 'CPG: uniform pruning ratio')
flags.DEFINE_boolean('cpg_skip_ht_layers', True, 'CPG: skip head & tail layers for pruning')
flags.DEFINE_string('cpg_prune_ratio_file', None, 'CPG: file path to the list of pruning ratios')
flags.DEFINE_float('cpg_lrn_rate_pgd_init', 1e-10, 'CPG: proximal gradient descent\'s initial learning rate')
flags.DEFINE_float('cpg_lrn_rate_pgd_incr', 1.4, 'CPG: proximal gradient descent\'s learning rate\'s increase ratio')
flags.DEFINE_float('cpg_lrn_rate_pgd_decr', 0.7, 'CPG: proximal gradient descent\'s learning rate\'s decrease ratio')
flags.DEFINE_float('cpg_lrn_rate_adam', 1e-2, 'CPG: Adam\'s initial learning rate')
flags.DEFINE_integer('cpg_nb_iters_layer', 1000, 'CPG: # of iterations for layer-wise FT')

log = logging.getLogger('pocketflow_amd')


def get_vars_by_scope(graph):
  """all / trainable / maskable (= kernels read by a Conv2D op, reference :51-71) variables of one network."""
  st = graph.store
  conv_kernels = {id(op.var) for op in graph.matmul_ops if op.type == 'Conv2D'}
  return {'all': list(st.vars), 'trainable': st.trainable_vars,
          'maskable': [v for v in st.trainable_vars if id(v) in conv_kernels]}


def proximal_shrink(w_krsc: torch.Tensor, prune_perctl: float):
  """W' [O, RS, I] -> (W' * max(1 - tau / n_c, 0), n_c) with n_c the L2 norm over (O, RS) of input channel c and
  tau = tf.contrib.distributions.percentile(n, prune_perctl) ('nearest')."""
  norm = torch.sqrt((w_krsc.float() ** 2).sum(dim=(0, 1)))                       # [I]
  n = norm.numel()
  q = np.float64(np.float32(prune_perctl))
  idx = int(np.clip(np.rint(np.float64(n - 1) * (np.float64(1.0) - q / np.float64(100.0))), 0, n - 1))
  threshold = torch.sort(norm, descending=True).values[idx]
  shrk = torch.clamp(1.0 - threshold / norm, min=0.0)                            # tau / 0 = inf -> 0
  shrk = torch.where(torch.isnan(shrk), torch.zeros_like(shrk), shrk)            # 0 / 0 (dead channel, tau = 0)
  return w_krsc * shrk.to(w_krsc.dtype), norm


def proximal_step(w_flat, g_flat, lrn_rate_pgd: float, prune_perctl: float, rows: int, cin: int, ws) -> None:
  """layer_ops[idx]['prune'] of the reference (:379-383) on the device, in place on the float32 master kernel `w_flat` (KRSC storage =
  [rows][cin]): pf_prox_norms (one read of W and G) -> nearest-rank percentile of the cin norms (pf_kth_largest_nonneg, the index in
  float64 on the float32 percentile as for the weight-sparsification masks) -> pf_prox_apply (one read of W and G, one write of W).
  Until round 4: `proximal_shrink` above, five torch ops."""
  partial, norms, thr, kth_ws = ws
  hip.prox_norms(w_flat, g_flat, float(lrn_rate_pgd), rows, cin, partial, norms)
  q = np.float64(np.float32(prune_perctl))
  idx = int(np.clip(np.rint(np.float64(cin - 1) * (np.float64(1.0) - q / np.float64(100.0))), 0, cin - 1))
  hip.kth_largest_nonneg(norms[:cin], idx, thr, kth_ws)
  hip.prox_apply(w_flat, g_flat, float(lrn_rate_pgd), rows, cin, norms, thr)


class ChannelPrunedGpuLearner(AbstractLearner):  # pylint: disable=too-many-instance-attributes
  """Channel pruning learner with GPU-based optimization."""

  def __init__(self, sm_writer, model_helper):
    super(ChannelPrunedGpuLearner, self).__init__(sm_writer, model_helper)
    self.model_scope_full = 'model'
    self.model_scope_prnd = 'pruned_model'
    if self.is_primary_worker('local'):
      self.download_model()  # pre-trained model is required
    self.auto_barrier()
    if FLAGS.enbl_dst:
      self.helper_dst = DistillationHelper(sm_writer, model_helper, self.mpi_comm)
    self.__build_train()
    self.__build_eval()

  # -- reference surface ----------------------------------------------------------------------------------------
  def train(self):
    """Choose channels layer by layer, then fine-tune the network with the chosen channels only."""
    save_path = checkpoint.latest_checkpoint(os.path.dirname(self.save_path_full))
    self.graph_full.store.load_numpy(checkpoint.load(save_path), strict=False)
    self.__init_pruned_model()
    if FLAGS.enbl_multi_gpu:
      self.bcast_op()

    self.__choose_channels()
    if self.is_primary_worker('global'):
      self.__save_model(is_train=True)
      self.evaluate()
    self.auto_barrier()

    nb_iters = FLAGS.nb_iters_override or self.nb_iters_train
    time_prev = timer()
    for idx_iter in range(nb_iters):
      log_rslt = self.train_step()
      if (idx_iter + 1) % FLAGS.summ_step == 0 and self.is_primary_worker('global'):
        self.__monitor_progress(log_rslt, idx_iter, timer() - time_prev)
        time_prev = timer()
      if self.is_primary_worker('global') and (idx_iter + 1) % FLAGS.save_step == 0:
        self.__save_model(is_train=True)
        self.evaluate()
      self.auto_barrier()

    rslt = None
    if self.is_primary_worker('global'):
      self.__save_model(is_train=True)
      self.__restore_model(is_train=False)
      self.__save_model(is_train=False)
      rslt = self.evaluate()
    return rslt

  def evaluate(self):
    """Restore a model from the latest checkpoint files and then evaluate it."""
    self.__restore_model(is_train=False)
    return self.run_eval()

  def run_eval(self):
    nb_iters = FLAGS.nb_eval_batches_override or int(np.ceil(float(FLAGS.nb_smpls_eval) / FLAGS.batch_size_eval))
    g = self.graph
    g.store.sync_compute()
    self.iter_eval.reset()
    pr_trn = calc_prune_ratio(self.vars_prnd['trainable'], self.device)
    pr_msk = calc_prune_ratio(self.vars_prnd['maskable'], self.device)
    rows, names = [], None
    self.dump_n_eval(outputs=None, action='init')
    with torch.no_grad():
      for __ in range(nb_iters):
        images, labels = self.iter_eval.get_next()
        x, y = self.to_device(images, labels)
        g.begin_step()
        with g.as_default():
          logits = self.forward_eval(x)
          loss, metrics = self.calc_loss(y, logits, self.vars_prnd['trainable'])
          if FLAGS.enbl_dst:
            loss = loss + self.helper_dst.calc_loss(logits, self.helper_dst.calc_logits(None, x))
        self.dump_n_eval(outputs=logits, action='dump')
        names = ['loss', 'pr_trn', 'pr_msk'] + list(metrics.keys())
        rows.append([float(loss), pr_trn, pr_msk] + [float(v) for v in metrics.values()])
    self.dump_n_eval(outputs=None, action='eval')
    means = np.mean(np.array(rows), axis=0)
    out = {}
    for idx, name in enumerate(names):
      log.info('%s = %.4e' % (name, means[idx]))
      out[name] = float(means[idx])
    return out

  def train_step(self):
    """train_op: fwd, loss (+ distillation), bwd, [all-reduce], grad * mask + Momentum in one fused launch."""
    g = self.graph
    g.store.sync_compute()
    images, labels = self.iter_train.get_next()
    x, y = self.to_device(images, labels)
    g.begin_step()
    with g.as_default():
      logits_dst = self.helper_dst.calc_logits(None, x) if FLAGS.enbl_dst else None
      logits = self.forward_train(x)
      loss, metrics = self.calc_loss(y, logits, self.vars_prnd['trainable'])
      if FLAGS.enbl_dst:
        loss = loss + self.helper_dst.calc_loss(logits, logits_dst)
    self.optimizer.backward(loss)
    lr = self.lrn_rate(self.global_step)
    self.optimizer.weight_decay = g.store.weight_decay
    self.optimizer.compute_gradients()
    self.optimizer.apply_gradients(lr)
    self.global_step += 1
    return [lr, float(loss.detach()), None, None] + [float(v) for v in metrics.values()], list(metrics.keys())

  # -- graphs ------------------------------------------------------------------------------------------------------
  def __build_train(self):
    self.graph_full = self.build_graph(self.model_scope_full, requires_grad=False)
    self.graph = self.build_graph(self.model_scope_prnd)
    self.save_path_full = FLAGS.save_path
    st = self.graph.store
    self.iter_train = self.build_dataset_train().to(self.device)
    self.vars_full = get_vars_by_scope(self.graph_full)
    self.vars_prnd = get_vars_by_scope(self.graph)
    self.maskable_var_names = [var.name for var in self.vars_prnd['maskable']]
    self.nb_layers = len(self.vars_prnd['maskable'])
    self.global_step = 0
    self.lrn_rate, self.nb_iters_train = self.setup_lrn_rate(self.global_step)
    self.masks = torch.ones_like(st.w_master)                   # all pruning masks, one flat buffer
    self.layer_mask = torch.zeros_like(st.w_master)              # mask restricted to the layer being re-fitted
    self.o_zero = torch.zeros_like(st.o_master)
    self.opt_layer = FlatOptimizer(st, 'adam')                   # layer_ops[i]['finetune'] (one slot set, restarted per layer)
    base = FlatOptimizer(st, 'momentum', momentum=FLAGS.momentum)
    base.w_mask = self.masks
    self.optimizer = base if not FLAGS.enbl_multi_gpu else mgw.DistributedOptimizer(base)
    if FLAGS.enbl_multi_gpu:
      self.opt_layer = mgw.DistributedOptimizer(self.opt_layer)
      self.bcast_op = mgw.broadcast_global_variables(0, [st], [self.optimizer, self.opt_layer])
    images, __ = self.iter_train.get_next()
    self.iter_train.reset()
    self.core_full = layers_of_vars(self.graph_full, self.forward_eval, images, self.vars_full['maskable'])
    self.core_prnd = layers_of_vars(self.graph, self.forward_eval, images, self.vars_prnd['maskable'])
    assert all(isinstance(l, Conv2D) for l in self.core_prnd)

  def __build_eval(self):
    self.iter_eval = self.build_dataset_eval().to(self.device)

  def __init_pruned_model(self):
    """init_op + init_opt_op: the channel-pruned model starts as a copy of the full model, step 0, empty slots."""
    st_f, st_p = self.graph_full.store, self.graph.store
    st_p.w_master.copy_(st_f.w_master)
    st_p.o_master.copy_(st_f.o_master)
    st_p.state.copy_(st_f.state)
    st_p.sync_compute()
    self.masks.fill_(1.0)
    self.global_step = 0
    self.optimizer.reset_slots()
    self.opt_layer.reset_slots()

  # -- channel selection ---------------------------------------------------------------------------------------------------
  def __regression_grad(self, idx):
    """reg_losses[idx] and its gradient w.r.t. kernel idx (left in the flat gradient buffer)."""
    images, __ = self.iter_train.get_next()
    layer_f, layer_p = self.core_full[idx], self.core_prnd[idx]
    st_f = self.graph_full.store
    state_full = st_f.state.clone()
    fuse_f, fuse_p = self.graph_full.fuse_conv1x1, self.graph.fuse_conv1x1
    self.graph_full.fuse_conv1x1 = self.graph.fuse_conv1x1 = False
    try:
      y_full = forward_tapped(self.graph_full, self.forward_train, images, layer_f, tap_dense=False, grad=True)[layer_f][1]
      st_f.state.copy_(state_full)                                  # the full model's BN statistics never move
      x_prnd = forward_tapped(self.graph, self.forward_train, images, layer_p, tap_dense=False, grad=True)[layer_p][0]
      diff = layer_p.plain(x_prnd.detach()).float() - y_full.detach().float()
      loss = (diff * diff).sum() / 2
      loss.backward()
    finally:
      self.graph_full.fuse_conv1x1, self.graph.fuse_conv1x1 = fuse_f, fuse_p
    st = self.graph.store
    if FLAGS.enbl_multi_gpu and dist.is_initialized() and dist.get_world_size() > 1:
      dist.all_reduce(st.w_grad, op=dist.ReduceOp.SUM)
      st.w_grad.div_(dist.get_world_size())
    return float(loss.detach())

  def __prune_step(self, idx, lrn_rate_pgd, prune_perctl):
    """layer_ops[idx]['prune']: one stochastic proximal gradient step on kernel idx; returns the regression loss."""
    reg_loss = self.__regression_grad(idx)
    st = self.graph.store
    var = self.vars_prnd['maskable'][idx]
    kh, kw, cin, cout = var.ref_shape
    sl = slice(var.offset, var.offset + var.numel)
    proximal_step(st.w_master[sl], st.w_grad[sl], lrn_rate_pgd, prune_perctl, cout * kh * kw, cin, self.__prox_ws(cout * kh * kw, cin))
    st.zero_grad()
    st.sync_compute()
    return reg_loss

  def __prox_ws(self, rows, cin):
    """(partial sums, norms, threshold, radix-select workspace) of the fused proximal step, grown on demand."""
    need = hip.prox_groups(rows, cin) * cin
    ws = getattr(self, '_prox_ws', None)
    if ws is None or ws[0].numel() < need or ws[1].numel() < cin:
      dev = self.graph.store.device
      ws = self._prox_ws = (torch.empty(max(need, 1 << 16), dtype=torch.float32, device=dev),
                            torch.empty(max(cin, 4096), dtype=torch.float32, device=dev),
                            torch.empty(1, dtype=torch.float32, device=dev), torch.empty(4096, dtype=torch.int32, device=dev))
    return ws

  def __update_mask(self, idx):
    """mask_updt_ops[idx]: mask = (||W[:, :, c, :]|| > 0) broadcast over the kernel (:243-250)."""
    st = self.graph.store
    var = self.vars_prnd['maskable'][idx]
    kh, kw, cin, cout = var.ref_shape
    sl = slice(var.offset, var.offset + var.numel)
    norm = torch.sqrt((st.w_master[sl].view(cout, kh * kw, cin).float() ** 2).sum(dim=(0, 1)))
    keep_in = (norm > 0.0).to(torch.uint8)
    keep_out = torch.ones(cout, dtype=torch.uint8, device=keep_in.device)
    hip.cp_build_mask(self.masks[sl], keep_in, keep_out, cout, kh * kw, cin)
    return int(keep_in.sum())

  def __finetune_step(self, idx):
    """layer_ops[idx]['finetune']: Adam on the regression loss w.r.t. kernel idx with masked gradients."""
    reg_loss = self.__regression_grad(idx)
    base = self.opt_layer.opt if FLAGS.enbl_multi_gpu else self.opt_layer
    base.g_scale, base.weight_decay = 1.0, 0.0
    base.apply_gradients(FLAGS.cpg_lrn_rate_adam)
    self.graph.store.sync_compute()
    return reg_loss

  def __choose_channels(self):  # pylint: disable=too-many-locals
    if FLAGS.cpg_prune_ratio_type == 'uniform':
      ratio_list = [FLAGS.cpg_prune_ratio] * self.nb_layers
      if FLAGS.cpg_skip_ht_layers:
        ratio_list[0] = 0.0
        ratio_list[-1] = 0.0
    elif FLAGS.cpg_prune_ratio_type == 'list':
      with open(FLAGS.cpg_prune_ratio_file, 'r') as i_file:
        ratio_list = [float(sub_str) for sub_str in i_file.readline().strip().split(',')]
    else:
      raise ValueError('unrecognized pruning ratio type: ' + FLAGS.cpg_prune_ratio_type)

    nb_workers = mgw.size() if FLAGS.enbl_multi_gpu else 1
    nb_iters_layer = max(int(FLAGS.cpg_nb_iters_layer / nb_workers), 1)
    primary = self.is_primary_worker('global')
    base_layer = self.opt_layer.opt if FLAGS.enbl_multi_gpu else self.opt_layer
    self.actual_prune_ratios = [0.0] * self.nb_layers
    for idx_layer in range(self.nb_layers):
      if ratio_list[idx_layer] == 0.0:
        continue
      var = self.vars_prnd['maskable'][idx_layer]
      if primary:
        log.info('layer #%d: pr = %.2f (target)' % (idx_layer, ratio_list[idx_layer]))
        log.info('mask.shape = {}'.format(var.ref_shape))
      time_prev = timer()
      reg_loss_prev = 0.0
      lrn_rate_pgd = FLAGS.cpg_lrn_rate_pgd_init
      for idx_iter in range(nb_iters_layer):
        prune_perctl = ratio_list[idx_layer] * 100.0 * (idx_iter + 1) / nb_iters_layer
        reg_loss = self.__prune_step(idx_layer, lrn_rate_pgd, prune_perctl)
        if primary and (idx_iter + 1) % max(nb_iters_layer // 10, 1) == 0:
          log.info('iter %d: loss = %.2e | lr = %.2e | percentile = %.2f' % (idx_iter + 1, reg_loss, lrn_rate_pgd, prune_perctl))
        lrn_rate_pgd *= FLAGS.cpg_lrn_rate_pgd_incr if reg_loss < reg_loss_prev else FLAGS.cpg_lrn_rate_pgd_decr
        reg_loss_prev = reg_loss

      # fine-tune with selected channels only
      nb_chns_nnz = self.__update_mask(idx_layer)
      sl = slice(var.offset, var.offset + var.numel)
      self.layer_mask.zero_()
      self.layer_mask[sl] = self.masks[sl]
      base_layer.w_mask, base_layer.o_mask = self.layer_mask, self.o_zero
      base_layer.reset_slots()
      for idx_iter in range(nb_iters_layer):
        reg_loss = self.__finetune_step(idx_layer)
        if primary and (idx_iter + 1) % max(nb_iters_layer // 10, 1) == 0:
          log.info('iter %d: nnz-chns = %d | loss = %.2e' % (idx_iter + 1, nb_chns_nnz, reg_loss))
      cin = var.ref_shape[2]
      self.actual_prune_ratios[idx_layer] = 1.0 - float(nb_chns_nnz) / cin
      if primary:
        log.info('layer #%d: pr = %.2f (actual) | time = %.2f' % (idx_layer, self.actual_prune_ratios[idx_layer], timer() - time_prev))
    if primary:
      log.info('pr_trn = %.4e | pr_msk = %.4e' % (calc_prune_ratio(self.vars_prnd['trainable'], self.device),
                                                  calc_prune_ratio(self.vars_prnd['maskable'], self.device)))

  # -- checkpoints / logging -------------------------------------------------------------------------------------------------
  def __save_model(self, is_train):
    if is_train:
      save_path = self.save_vars(FLAGS.cpg_save_path, self.global_step)
    else:
      save_path = self.save_vars(FLAGS.cpg_save_path_eval)
    log.info('model saved to ' + save_path)

  def __restore_model(self, is_train):
    save_path = checkpoint.latest_checkpoint(os.path.dirname(FLAGS.cpg_save_path))
    self.restore_vars(save_path)
    log.info('model restored from ' + save_path)

  def __monitor_progress(self, log_rslt, idx_iter, time_step):
    vals, metric_names = log_rslt
    vals[2] = calc_prune_ratio(self.vars_prnd['trainable'], self.device)
    vals[3] = calc_prune_ratio(self.vars_prnd['maskable'], self.device)
    names = ['lr', 'loss', 'pr_trn', 'pr_msk'] + metric_names
    if self.sm_writer is not None:
      self.sm_writer.add_summary(dict(zip(names, vals)), idx_iter)
    speed = FLAGS.batch_size * FLAGS.summ_step / time_step
    if FLAGS.enbl_multi_gpu:
      speed *= mgw.size()
    log_str = ' | '.join(['%s = %.4e' % (name, value) for name, value in zip(names, vals)])
    log.info('iter #%d: %s | speed = %.2f pics / sec' % (idx_iter + 1, log_str, speed))
    self.last_speed = speed
