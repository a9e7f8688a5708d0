"""Weight sparsification learner (reference learners/weight_sparsification/learner.py:32-375).

Dynamic magnitude pruning on the Zhu-Gupta schedule: Momentum fine-tuning with `grad * mask` fused into
the optimiser kernel; every `ws_mask_update_step` steps inside [ws_iter_ratio_beg, ws_iter_ratio_end]
of training (and once after) every maskable kernel goes through the mask refresh
    bkup <- where(mask > .5, var, bkup); thr <- percentile(|bkup|, r_t * 100);
    mask <- |bkup| > thr; var <- bkup * mask
executed as: fused merge+abs kernel -> 4-pass radix select of the k-th |w| (no sort) -> fused
mask/apply kernel; then the Momentum slots are re-initialised (:124-131, 283-288).
"""
from __future__ import annotations

import logging
import os
from timeit import default_timer as timer

import numpy as np
import torch

from pocketflow_amd import hip
from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.learners.abstract_learner import AbstractLearner
from pocketflow_amd.learners.distillation_helper import DistillationHelper
from pocketflow_amd.learners import teacher_ahead
from pocketflow_amd.learners.weight_sparsification.pr_optimizer import PROptimizer
from pocketflow_amd.learners.weight_sparsification.utils import get_maskable_vars
from pocketflow_amd.optim import FlatOptimizer
from pocketflow_amd.utils import checkpoint
from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw

flags.DEFINE_string('ws_save_path', './models_ws/model.ckpt', 'WS: model\'s save path')
flags.DEFINE_float('ws_prune_ratio', 0.75, 'WS: target pruning ratio')
flags.DEFINE_string('ws_prune_ratio_prtl', 'optimal', 'WS: pruning ratio protocol (\'uniform\' | \'heurist\' | \'optimal\')')
flags.DEFINE_integer('ws_nb_rlouts', 200, 'WS: # of roll-outs for the RL agent')
flags.DEFINE_integer('ws_nb_rlouts_min', 50, 'WS: minimal # of roll-outs for the RL agent to start training')
flags.DEFINE_string('ws_reward_type', 'single-obj', 'WS: reward type (\'single-obj\' OR \'multi-obj\')')
flags.DEFINE_float('ws_lrn_rate_rg', 3e-2, 'WS: learning rate for layerwise regression')
flags.DEFINE_integer('ws_nb_iters_rg', 20, 'WS: # of iterations for layerwise regression')
flags.DEFINE_float('ws_lrn_rate_ft', 3e-4, 'WS: learning rate for global fine-tuning')
flags.DEFINE_integer('ws_nb_iters_ft', 400, 'WS: # of iterations for global fine-tuning')
flags.DEFINE_integer('ws_nb_iters_feval', 25, 'WS: # of iterations for fast evaluation')
flags.DEFINE_float('ws_prune_ratio_exp', 3.0, 'WS: pruning ratio\'s exponent term')
flags.DEFINE_float('ws_iter_ratio_beg', 0.1, 'WS: iteration ratio (at starting time)')
flags.DEFINE_float('ws_iter_ratio_end', 0.5, 'WS: iteration ratio (at ending time)')
flags.DEFINE_float('ws_mask_update_step', 500, 'WS: step size for updating the pruning mask')

log = logging.getLogger('pocketflow_amd')


def calc_prune_ratio(vars_list, device):
  """Overall pruning ratio of the given variables: 1 - count_nonzero / size (reference :51-65)."""
  cnt = torch.zeros(1, dtype=torch.int64, device=device)
  nb_params_all = 0
  for var in vars_list:
    hip.count_nonzero(var.master.reshape(-1), cnt)
    nb_params_all += var.numel
  nnz = np.float32(int(cnt.item()))
  return float(np.float32(1.0) - nnz / np.float32(nb_params_all))


class WeightSparseLearner(AbstractLearner):  # pylint: disable=too-many-instance-attributes
  """Weight sparsification learner."""

  def __init__(self, sm_writer, model_helper):
    super(WeightSparseLearner, self).__init__(sm_writer, model_helper)
    self.mask_scope = 'mask'

    if FLAGS.exec_mode == 'train':
      pr_optimizer = PROptimizer(model_helper, self.mpi_comm)
      self.var_names_n_prune_ratios = pr_optimizer.run()

    if FLAGS.enbl_dst:
      self.helper_dst = DistillationHelper(sm_writer, model_helper, self.mpi_comm)
    self.__build_train()
    self.__build_eval()

  # ---------------------------------------------------------------------------------------------
  def _train_step_eager(self):
    """`sess.run(train_op)`: fwd, loss, bwd, [all-reduce], grad*mask + Momentum (one fused launch)."""
    g = self.graph
    g.store.sync_compute()
    ahead, x, y, logits_dst = teacher_ahead.next_batch(self)   # batch (+ teacher logits issued by the previous step on the side stream)
    g.begin_step()
    with g.as_default():
      if FLAGS.enbl_dst and logits_dst is None:
        logits_dst = self.helper_dst.calc_logits(None, x)
      logits = self.forward_train(x)
      loss, metrics = self.calc_loss(y, logits, self.trainable_vars)
      if FLAGS.enbl_dst:
        loss = loss + self.helper_dst.calc_loss(logits, logits_dst)
    self.optimizer.backward(loss)
    lr = self.lrn_rate(self.global_step)
    self.optimizer.weight_decay = g.store.weight_decay
    self.optimizer.compute_gradients()
    self.optimizer.apply_gradients(lr)
    self.global_step += 1
    if ahead is not None:
      ahead.issue()                                 # next batch's teacher forward on the side stream: it runs beside the NEXT step's forward pass
    return lr, loss, metrics

  def prune_step(self):
    """[prune_op, init_opt_op]: refresh every maskable variable's mask, re-initialise Momentum slots."""
    st = self.graph.store
    for var, (name, prune_ratio_fnl) in zip(self.maskable_vars, self.var_names_n_prune_ratios):
      assert var.name == name, 'unmatched variable names: %s vs. %s' % (var.name, name)
      r_t = self.__calc_prune_ratio_dyn(prune_ratio_fnl)
      n = var.numel
      sl = slice(var.offset, var.offset + n)
      v, b, m = st.w_master[sl], self.var_bkup[sl], self.masks[sl]
      hip.ws_bkup_merge_abs(v, b, m, self.abs_buf[:n])
      # tf.contrib.distributions.percentile(|bkup|, r_t * 100), 'nearest': descending sort, index
      # round_half_even((n - 1) * (1 - q / 100)) evaluated in float64 on the float32 ratio * 100
      q = np.float64(np.float32(np.float32(r_t) * np.float32(100.0)))
      idx = int(np.clip(np.rint(np.float64(n - 1) * (np.float64(1.0) - q / np.float64(100.0))), 0, n - 1))
      hip.kth_largest_nonneg(self.abs_buf[:n], idx, self.thr, self.kth_ws)
      hip.ws_mask_apply(v, b, m, self.thr)
    self.optimizer.reset_slots()

  def train(self):
    """Train a model and periodically produce checkpoint files."""
    if FLAGS.enbl_multi_gpu:
      self.bcast_op()
    nb_iters = FLAGS.nb_iters_override or self.nb_iters_train
    last_mask_applied = False
    time_prev = timer()
    for idx_iter in range(nb_iters):
      lr, loss, metrics = self.train_step()
      if (idx_iter + 1) % FLAGS.summ_step == 0 and self.is_primary_worker('global'):
        torch.cuda.synchronize()
        time_step = timer() - time_prev
        self.__monitor_progress(lr, loss, metrics, idx_iter, time_step)
        time_prev = timer()

      # apply pruning
      if (idx_iter + 1) % FLAGS.ws_mask_update_step == 0:
        iter_ratio = float(idx_iter + 1) / self.nb_iters_train
        if iter_ratio >= FLAGS.ws_iter_ratio_beg:
          if iter_ratio <= FLAGS.ws_iter_ratio_end:
            self.prune_step()
          elif not last_mask_applied:
            last_mask_applied = True
            self.prune_step()

      if self.is_primary_worker('global') and (idx_iter + 1) % FLAGS.save_step == 0:
        self.__save_model()
        self.evaluate()

    if self.is_primary_worker('global'):
      self.__save_model()
      return self.evaluate()
    return None

  def evaluate(self):
    """Restore a model from the latest checkpoint files and then evaluate it."""
    self.__restore_model(is_train=False)
    return self.run_eval()

  def run_eval(self):
    nb_iters = FLAGS.nb_eval_batches_override or int(np.ceil(float(FLAGS.nb_smpls_eval) / FLAGS.batch_size_eval))
    g = self.graph
    g.store.sync_compute()
    self.iter_eval.reset()
    pr_trn = calc_prune_ratio(self.trainable_vars, self.device)
    pr_msk = calc_prune_ratio(self.maskable_vars, self.device)
    rslts, names = [], None
    with torch.no_grad():
      for __ in range(nb_iters):
        images, labels = self.iter_eval.get_next()
        x, y = self.to_device(images, labels)
        g.begin_step()
        with g.as_default():
          logits = self.forward_eval(x)
          loss, metrics = self.calc_loss(y, logits, self.trainable_vars)
          if FLAGS.enbl_dst:
            loss = loss + self.helper_dst.calc_loss(logits, self.helper_dst.calc_logits(None, x))
        rslts.append([float(loss), pr_trn, pr_msk] + [float(v) for v in metrics.values()])
        names = ['loss', 'pr_trn', 'pr_msk'] + list(metrics.keys())
    means = np.mean(np.array(rslts), axis=0)
    out = {}
    for idx, name in enumerate(names):
      log.info('%s = %.4e' % (name, means[idx]))
      out[name] = float(means[idx])
    return out

  # ---------------------------------------------------------------------------------------------
  def __build_train(self):  # pylint: disable=too-many-locals
    self.graph = self.build_graph(self.model_scope)
    st = self.graph.store
    self.iter_train = self.build_dataset_train().to(self.device)
    self.maskable_vars = get_maskable_vars(self.trainable_vars)
    self.maskable_var_names = [var.name for var in self.maskable_vars]
    assert all(v.group == 'W' for v in self.maskable_vars)
    self.global_step = 0
    self.lrn_rate, self.nb_iters_train = self.setup_lrn_rate(self.global_step)
    self.__build_masks()
    optimizer_base = FlatOptimizer(st, 'momentum', momentum=FLAGS.momentum)
    optimizer_base.w_mask = self.masks
    self.optimizer = optimizer_base if not FLAGS.enbl_multi_gpu else mgw.DistributedOptimizer(optimizer_base)
    if FLAGS.enbl_multi_gpu:
      self.bcast_op = mgw.broadcast_global_variables(0, [st], [self.optimizer])

  def __build_eval(self):
    self.iter_eval = self.build_dataset_eval().to(self.device)

  def __build_masks(self):
    """mask = ones, var_bkup = initial value for every maskable variable (:277-280); kept as flat
    buffers parallel to the matmul-kernel master buffer (non-maskable kernels keep mask == 1)."""
    st = self.graph.store
    self.masks = torch.ones_like(st.w_master)
    self.var_bkup = st.w_master.clone()
    max_n = max([v.numel for v in self.maskable_vars] + [1])
    self.abs_buf = torch.empty(max_n, dtype=torch.float32, device=self.device)
    self.thr = torch.empty(1, dtype=torch.float32, device=self.device)
    self.kth_ws = torch.empty(1024, dtype=torch.int32, device=self.device)

  def __calc_prune_ratio_dyn(self, prune_ratio_fnl):
    """Dynamic pruning ratio r_f * (1 - (1 - clip((step - t_b) / (t_e - t_b), 0, 1)) ** exp), float32."""
    idx_iter_beg = int(self.nb_iters_train * FLAGS.ws_iter_ratio_beg)
    idx_iter_end = int(self.nb_iters_train * FLAGS.ws_iter_ratio_end)
    base = np.float32(np.float32(self.global_step - idx_iter_beg) / np.float32(idx_iter_end - idx_iter_beg))
    base = np.minimum(np.float32(1.0), np.maximum(np.float32(0.0), base))
    one = np.float32(1.0)
    return np.float32(np.float32(prune_ratio_fnl) *
                      (one - np.float32(np.power(one - base, np.float32(FLAGS.ws_prune_ratio_exp)))))

  def __save_model(self):
    save_path = self.save_vars(FLAGS.ws_save_path, self.global_step)
    log.info('model saved to ' + save_path)

  def __restore_model(self, is_train):
    save_path = checkpoint.latest_checkpoint(os.path.dirname(FLAGS.ws_save_path))
    self.restore_vars(save_path)
    log.info('model restored from ' + save_path)

  def __monitor_progress(self, lr, loss, metrics, idx_iter, time_step):
    speed = FLAGS.batch_size * FLAGS.summ_step / time_step
    if FLAGS.enbl_multi_gpu:
      speed *= mgw.size()
    pr_trn = calc_prune_ratio(self.trainable_vars, self.device)
    pr_msk = calc_prune_ratio(self.maskable_vars, self.device)
    names = ['lr', 'loss', 'pr_trn', 'pr_msk'] + list(metrics.keys())
    vals = [lr, float(loss.detach()), pr_trn, pr_msk] + [float(v) for v in metrics.values()]
    if self.sm_writer is not None:
      self.sm_writer.add_summary(dict(zip(names, vals)), idx_iter)
    log_str = ' | '.join(['%s = %.4e' % (n, v) for n, v in zip(names, vals)])
    log.info('iter #%d: %s | speed = %.2f pics / sec' % (idx_iter + 1, log_str, speed))
