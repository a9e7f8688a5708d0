"""Pruning-ratio optimizer of the weight-sparsification learner (reference learners/weight_sparsification/pr_optimizer.py:96-611).

`ws_prune_ratio_prtl`:
  uniform  every maskable tensor gets ws_prune_ratio                                          (:385-392)
  heurist  ratio_i = alpha * log(n_i), alpha s.t. the overall ratio is ws_prune_ratio          (:394-409)
  optimal  a DDPG agent proposes one ratio per tensor; the proposal is scored by a short re-training of the
           magnitude-pruned network and the best of ws_nb_rlouts proposals wins                (:411-611)

One roll-out of `optimal` (= the hot path again, on inference-mode networks, `forward_eval` as in the reference):
  1. masks  m_i = |w_i| > percentile(|w_i|, r_i * 100) from the FULL network's weights; pruned net <- full * mask
     (device: the mask-refresh kernels of the learner -- merge/abs, radix select of the k-th |w|, mask/apply);
  2. layer-wise regression: for every maskable tensor, ws_nb_iters_rg Adam(ws_lrn_rate_rg) steps on
     l2_loss(conv_i(pruned net) - conv_i(full net)) w.r.t. that tensor only, masked;
  3. ws_nb_iters_ft Adam(ws_lrn_rate_ft) steps on the task loss over all trainables of the pruned net, masked;
  4. reward = accuracy (top-5 on ILSVRC-12) over ws_nb_iters_feval batches of the validation split.
Both networks live in one process as two `Graph`s (scopes `model` / `pruned_model`); step 2 runs both forward
passes in tap mode only up to the layer being regressed and back-propagates through that single convolution.

Control-plane differences from the reference: ratios / rewards reach the other ranks through `mpi_comm.bcast`
instead of ./ws.prune.ratios and ./ws.reward; the evaluation network IS the pruned training network (the reference
round-trips through a checkpoint into a second TF graph), the checkpoint under `models_pruned` is still written.
"""
from __future__ import annotations

import logging
import math
import os
from timeit import default_timer as timer

import numpy as np
import torch

from pocketflow_amd import hip
from pocketflow_amd.flags import FLAGS
from pocketflow_amd.learners.layerwise import forward_tapped, layers_of_vars
from pocketflow_amd.learners.weight_sparsification.rl_helper import RLHelper
from pocketflow_amd.learners.weight_sparsification.utils import get_maskable_vars
from pocketflow_amd.rl_agents.ddpg.agent import Agent as DdpgAgent
from pocketflow_amd.utils import checkpoint
from pocketflow_amd.utils.misc_utils import is_primary_worker
from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw

log = logging.getLogger('pocketflow_amd')


def get_vars_by_scope(graph):
  """all / trainable / maskable variables of one network (reference :32-47 filters collections by scope)."""
  st = graph.store
  return {'all': list(st.vars), 'trainable': st.trainable_vars, 'maskable': get_maskable_vars(st.trainable_vars)}


def percentile_index(n, ratio):
  """Index into the DESCENDING sort that tf.contrib.distributions.percentile(x, ratio * 100) ('nearest') returns:
  round_half_even((n - 1) * (1 - q / 100)) in float64 on the float32 product q = ratio * 100."""
  q = np.float64(np.float32(np.float32(ratio) * np.float32(100.0)))
  return int(np.clip(np.rint(np.float64(n - 1) * (np.float64(1.0) - q / np.float64(100.0))), 0, n - 1))


class PROptimizer(object):  # pylint: disable=too-many-instance-attributes
  """Pruning ratio optimizer for the weight sparsification learner."""

  def __init__(self, model_helper, mpi_comm):
    self.model_name = model_helper.model_name
    self.dataset_name = model_helper.dataset_name
    self.mpi_comm = mpi_comm
    self.data_scope = 'data'
    self.model_scope_full = 'model'
    self.model_scope_prnd = 'pruned_model'
    if FLAGS.ws_prune_ratio_prtl in ['uniform', 'heurist']:
      self.__build_minimal(model_helper)  # no RL-related state
    elif FLAGS.ws_prune_ratio_prtl == 'optimal':
      self.__build_train(model_helper)
      self.__build_eval(model_helper)
    else:
      raise ValueError('unrecognzed WS pruning ratio protocol: ' + FLAGS.ws_prune_ratio_prtl)

  def run(self):
    """Return the list of (variable name, pruning ratio) pairs of all maskable variables."""
    if FLAGS.ws_prune_ratio_prtl == 'uniform':
      var_names_n_prune_ratios = self.__calc_uniform_prune_ratios()
    elif FLAGS.ws_prune_ratio_prtl == 'heurist':
      var_names_n_prune_ratios = self.__calc_heurist_prune_ratios()
    else:
      var_names_n_prune_ratios = self.__calc_optimal_prune_ratios()
    if is_primary_worker('global'):
      for var_name, prune_ratio in var_names_n_prune_ratios:
        log.info('%s: %f' % (var_name, prune_ratio))
    return var_names_n_prune_ratios

  # -- graphs ---------------------------------------------------------------------------------------------
  def __build_minimal(self, model_helper):
    """Declare the model on a CPU-side graph only to enumerate its variables (no device work)."""
    from pocketflow_amd.graph import Graph
    from pocketflow_amd.learners.abstract_learner import input_spec
    graph = Graph(self.model_scope_full, 'cpu', torch.float32)
    with graph.as_default():
      model_helper.forward_train(input_spec(model_helper))
    self.vars_full = {'maskable': get_maskable_vars(graph.store.trainable_vars)}

  def __declare(self, model_helper, scope, requires_grad):
    from pocketflow_amd.graph import Graph
    from pocketflow_amd.learners.abstract_learner import compute_dtype, input_spec
    graph = Graph(scope, self.device, compute_dtype())
    # plain convolutions + the BN kernels: the fused 1x1 path is tuned for training-mode statistics
    graph.fuse_conv1x1 = False
    with graph.as_default():
      model_helper.forward_eval(input_spec(model_helper))  # DO NOT USE forward_train() HERE (reference :169)
    graph.finalize(separate_compute=False, seed=FLAGS.init_seed, requires_grad=requires_grad)
    return graph

  def __build_train(self, model_helper):  # pylint: disable=too-many-locals
    from pocketflow_amd.learners.abstract_learner import require_gpu
    from pocketflow_amd.optim import FlatOptimizer
    self.device = require_gpu()
    self.model_helper = model_helper
    self.iter_trn, self.iter_val = model_helper.build_dataset_train(enbl_trn_val_split=True)
    self.iter_trn.to(self.device)
    self.iter_val.to(self.device)

    # full-precision network (constant) and weight sparsified network (re-trained every roll-out)
    self.graph_full = self.__declare(model_helper, self.model_scope_full, requires_grad=False)
    self.graph_prnd = self.__declare(model_helper, self.model_scope_prnd, requires_grad=True)
    self.vars_full = get_vars_by_scope(self.graph_full)
    self.vars_prnd = get_vars_by_scope(self.graph_prnd)
    self.maskable_var_names = [var.name for var in self.vars_prnd['maskable']]
    self.save_path_full = FLAGS.save_path
    self.save_path_prnd = FLAGS.save_path.replace('models', 'models_pruned')
    assert all(v.group == 'W' for v in self.vars_prnd['maskable'])

    # pruning masks: one flat {0,1} buffer parallel to the matmul-kernel buffer (non-maskable kernels stay 1)
    st = self.graph_prnd.store
    self.masks = torch.ones_like(st.w_master)
    max_n = max([v.numel for v in self.vars_prnd['maskable']] + [1])
    self.abs_buf = torch.empty(max_n, dtype=torch.float32, device=self.device)
    self.bkup_buf = torch.empty(max_n, dtype=torch.float32, device=self.device)
    self.thr = torch.empty(1, dtype=torch.float32, device=self.device)
    self.kth_ws = torch.empty(1024, dtype=torch.int32, device=self.device)
    self.rg_mask = torch.zeros_like(st.w_master)       # masks restricted to the tensor being regressed

    # layer-wise regression & network fine-tuning optimisers (one Adam slot set each: only the tensor being
    # regressed sees non-zero gradients, so per-layer optimisers of the reference are this one with its
    # beta powers restarted per layer)
    self.opt_rg = FlatOptimizer(st, 'adam')
    self.opt_ft = FlatOptimizer(st, 'adam')
    self.opt_ft.w_mask = self.masks
    if FLAGS.enbl_multi_gpu:
      self.opt_rg, self.opt_ft = mgw.DistributedOptimizer(self.opt_rg), mgw.DistributedOptimizer(self.opt_ft)
      self.bcast_op = mgw.broadcast_global_variables(0, [st], [self.opt_rg, self.opt_ft])
    self.__find_core_layers()

    if is_primary_worker('global'):
      self.rl_helper, self.agent = self.__build_rl_helper_n_agent()

  def __build_eval(self, model_helper):
    """The evaluation network of the reference is a second TF graph restored from the pruned checkpoint; here
    the pruned network evaluates itself on the validation split (no_grad)."""
    self.metrics_eval = None

  def __find_core_layers(self):
    """Layer objects of both networks whose kernels are the maskable variables, in maskable order (the reference
    pairs ops matched by name patterns with `vars_prnd['maskable'][idx]`, :301-313)."""
    images, __ = self.iter_trn.get_next()
    self.iter_trn.reset()
    self.core_full = self.__layers_by_var(self.graph_full, self.vars_full['maskable'], images)
    self.core_prnd = self.__layers_by_var(self.graph_prnd, self.vars_prnd['maskable'], images)

  def __layers_by_var(self, graph, maskable, images):
    return layers_of_vars(graph, self.model_helper.forward_eval, images, maskable)

  def __forward_tapped(self, graph, images, stop_layer):
    """Inference forward pass in tap mode (no gradients) up to `stop_layer` (learners/layerwise.py)."""
    return forward_tapped(graph, self.model_helper.forward_eval, images, stop_layer)

  def __build_rl_helper_n_agent(self):
    skip_head_n_tail = (self.dataset_name == 'cifar_10')  # skip head & tail layers on CIFAR-10
    rl_helper = RLHelper(None, self.vars_full['maskable'], skip_head_n_tail)
    buf_size = len(self.vars_full['maskable']) * FLAGS.ws_nb_rlouts_min
    agent = DdpgAgent(None, rl_helper.s_dims, 1, FLAGS.ws_nb_rlouts, buf_size, 0.0, 1.0)
    return rl_helper, agent

  # -- closed-form protocols ----------------------------------------------------------------------------------
  def __calc_uniform_prune_ratios(self):
    return [(var.name, FLAGS.ws_prune_ratio) for var in self.vars_full['maskable']]

  def __calc_heurist_prune_ratios(self):
    nb_params = np.array([var.numel for var in self.vars_full['maskable']])
    alpha = FLAGS.ws_prune_ratio * np.sum(nb_params) / np.sum(nb_params * np.log(nb_params))
    return [(var.name, alpha * np.log(nb_params[idx])) for idx, var in enumerate(self.vars_full['maskable'])]

  # -- the search -------------------------------------------------------------------------------------------------
  def __bcast(self, obj):
    if FLAGS.enbl_multi_gpu and self.mpi_comm is not None:
      return self.mpi_comm.bcast(obj, root=0)
    return obj

  def __calc_optimal_prune_ratios(self):
    save_path = checkpoint.latest_checkpoint(os.path.dirname(self.save_path_full))
    self.graph_full.store.load_numpy(checkpoint.load(save_path), strict=False)
    primary = is_primary_worker('global')
    if primary:
      self.agent.init()
    reward_best = -np.inf
    prune_ratios_best = None
    self.reward_history = []
    for idx_rlout in range(FLAGS.ws_nb_rlouts):
      prune_ratios, states_n_actions = None, None
      if primary:
        log.info('starting %d-th roll-out' % idx_rlout)
        prune_ratios, states_n_actions = self.__calc_rlout_actions()
      # the reference passes the ratios through a '%f'-formatted text file: six decimals survive
      prune_ratios = np.array([float('%f' % r) for r in self.__bcast(prune_ratios)])

      reward = self.__calc_rlout_reward(prune_ratios)
      reward = float('%f' % self.__bcast(reward))
      self.reward_history.append(reward)

      if primary:
        self.agent.finalize_rlout(reward * np.ones(len(self.vars_full['maskable'])))
        self.__record_rlout_transitions(states_n_actions, reward)

      if reward_best < reward:
        if primary:
          log.info('best reward updated: %.4f -> %.4f' % (reward_best, reward))
          log.info('optimal pruning ratios: ' + ' '.join(['%.2f' % prune_ratio for prune_ratio in prune_ratios[:]]))
        reward_best = reward
        prune_ratios_best = np.copy(prune_ratios)

    return [(var_full.name, prune_ratios_best[idx]) for idx, var_full in enumerate(self.vars_full['maskable'])]

  def __calc_rlout_actions(self):
    """One ratio per maskable tensor from the noisy actor; the agent trains once per decision (:474-494)."""
    self.agent.init_rlout()
    prune_ratios, states_n_actions = [], []
    for idx in range(len(self.vars_full['maskable'])):
      state = self.rl_helper.calc_state(idx)
      action = self.agent.actions_noisy(state)
      prune_ratios += [self.rl_helper.cvt_action_to_prune_ratio(idx, action[0][0])]
      states_n_actions += [(state, action)]
      actor_loss, critic_loss, noise_std = self.agent.train()
    log.info('a-loss = %.2e | c-loss = %.2e | noise std. = %.2e' % (actor_loss, critic_loss, noise_std))
    return prune_ratios, states_n_actions

  def __init_pruned_network(self, prune_ratios):
    """pr_assign_op + init_op (:202-210, 254-281): masks from the full network, pruned <- full * mask elsewhere a copy."""
    st_f, st_p = self.graph_full.store, self.graph_prnd.store
    st_p.w_master.copy_(st_f.w_master)
    st_p.o_master.copy_(st_f.o_master)
    st_p.state.copy_(st_f.state)
    self.masks.fill_(1.0)
    for var, ratio in zip(self.vars_prnd['maskable'], prune_ratios):
      n = var.numel
      sl = slice(var.offset, var.offset + n)
      v, b, m = st_p.w_master[sl], self.bkup_buf[:n], self.masks[sl]
      hip.ws_bkup_merge_abs(v, b, m, self.abs_buf[:n])           # mask == 1: bkup <- var, abs <- |var|
      hip.kth_largest_nonneg(self.abs_buf[:n], percentile_index(n, ratio), self.thr, self.kth_ws)
      hip.ws_mask_apply(v, b, m, self.thr)                       # mask <- |bkup| > thr; var <- bkup * mask
    st_p.sync_compute()
    self.opt_rg.reset_slots()                          # rg_init_op
    self.opt_ft.reset_slots()                          # ft_init_op

  def __calc_rlout_reward(self, prune_ratios):
    self.__init_pruned_network(prune_ratios)
    if FLAGS.enbl_multi_gpu:
      self.bcast_op()
    primary = is_primary_worker('global')

    if primary:
      loss_pre, metrics_pre = self.__calc_loss_n_metrics()
      assert 'accuracy' in metrics_pre or 'acc_top5' in metrics_pre, \
        'either <accuracy> or <acc_top5> must be evaluated and returned'

    self.__retrain_network()

    reward = None
    if primary:
      loss_post, metrics_post = self.__calc_loss_n_metrics(save=True)
      key = 'accuracy' if 'accuracy' in metrics_post else 'acc_top5'
      reward_pre = self.rl_helper.calc_reward(metrics_pre[key])
      reward = self.rl_helper.calc_reward(metrics_post[key])
      prune_ratio = self.rl_helper.calc_overall_prune_ratio()
      metrics_diff = ' | '.join(['%s: %.4f -> %.4f' % (k, metrics_pre[k], metrics_post[k]) for k in metrics_post])
      log.info('loss: %.4e -> %.4e | %s | reward: %.4f -> %.4f | prune_ratio = %.4f'
               % (loss_pre, loss_post, metrics_diff, reward_pre, reward, prune_ratio))
    return reward

  # -- re-training --------------------------------------------------------------------------------------------------
  def __regression_step(self, idx):
    """One step of rg_train_ops[idx] (:283-316): Adam on l2_loss(conv_idx(pruned) - conv_idx(full)) w.r.t. kernel idx."""
    images, __ = self.iter_trn.get_next()
    layer_f, layer_p = self.core_full[idx], self.core_prnd[idx]
    y_full = self.__forward_tapped(self.graph_full, images, layer_f)[layer_f][1]
    x_prnd = self.__forward_tapped(self.graph_prnd, images, layer_p)[layer_p][0]
    diff = layer_p.plain(x_prnd.detach()).float() - y_full.float()
    loss = (diff * diff).sum() / 2
    self.opt_rg.backward(loss)
    self.opt_rg.weight_decay = 0.0
    self.opt_rg.compute_gradients()
    self.opt_rg.apply_gradients(FLAGS.ws_lrn_rate_rg)
    return loss

  def __finetune_step(self):
    """One step of ft_train_op (:318-337): Adam on the task loss over every trainable of the pruned net, masked."""
    from pocketflow_amd.graph import to_device_images
    g = self.graph_prnd
    images, labels = self.iter_trn.get_next()
    g.begin_step()
    with g.as_default():
      logits = self.model_helper.forward_eval(to_device_images(images, g))
      loss, __ = self.model_helper.calc_loss(labels.to(self.device), logits, self.vars_prnd['trainable'])
    self.opt_ft.backward(loss)
    self.opt_ft.weight_decay = g.store.weight_decay
    self.opt_ft.compute_gradients()
    self.opt_ft.apply_gradients(FLAGS.ws_lrn_rate_ft)
    return loss

  def __retrain_network(self):
    nb_workers = mgw.size() if FLAGS.enbl_multi_gpu else 1
    nb_iters_rg = int(math.ceil(FLAGS.ws_nb_iters_rg / nb_workers))
    nb_iters_ft = int(math.ceil(FLAGS.ws_nb_iters_ft / nb_workers))
    st = self.graph_prnd.store
    base_rg = self.opt_rg.opt if FLAGS.enbl_multi_gpu else self.opt_rg

    time_prev = timer()
    for idx, var in enumerate(self.vars_prnd['maskable']):
      sl = slice(var.offset, var.offset + var.numel)
      self.rg_mask.zero_()
      self.rg_mask[sl] = self.masks[sl]
      base_rg.w_mask = self.rg_mask
      base_rg.o_mask = self.__zero_o_mask()
      # the reference builds a SEPARATE AdamOptimizer per layer (pr_optimizer.py:283-316): fresh m / v / beta powers,
      # so that layers regressed earlier do not keep coasting on stale momentum while later layers are regressed
      base_rg.reset_slots()
      for __ in range(nb_iters_rg):
        self.__regression_step(idx)
      st.sync_compute()
    time_rg = timer() - time_prev

    time_prev = timer()
    for __ in range(nb_iters_ft):
      self.__finetune_step()
    st.sync_compute()
    time_ft = timer() - time_prev
    log.info('time consumption: %.4f (s) - RG | %.4f (s) - FT' % (time_rg, time_ft))

  def __zero_o_mask(self):
    if getattr(self, '_o_zero', None) is None:
      self._o_zero = torch.zeros_like(self.graph_prnd.store.o_master)
    return self._o_zero

  def __record_rlout_transitions(self, states_n_actions, reward):
    for idx, (state, action) in enumerate(states_n_actions):
      last = idx == len(states_n_actions) - 1
      terminal = np.ones((1, 1)) if last else np.zeros((1, 1))
      state_next = np.zeros_like(state) if last else states_n_actions[idx + 1][0]
      self.agent.record(state, action, reward * np.ones((1, 1)), terminal, state_next)

  def __calc_loss_n_metrics(self, save=False):
    """Loss & metrics of the pruned network on the validation split (:581-611).  `save`: also write the
    `models_pruned` checkpoint (the reference writes it before every evaluation because its evaluation graph is
    restored from it; here the network evaluates itself, so only the re-trained state of each roll-out is kept)."""
    from pocketflow_amd.graph import to_device_images
    g = self.graph_prnd
    if save:
      checkpoint.save(g.store.export_numpy(), self.save_path_prnd, None, fmt=FLAGS.ckpt_format)
    nb_iters = FLAGS.ws_nb_iters_feval if FLAGS.ws_nb_iters_feval > 0 else FLAGS.nb_smpls_eval // FLAGS.batch_size_eval
    rows, names = [], None
    with torch.no_grad():
      for __ in range(nb_iters):
        images, labels = self.iter_val.get_next()
        g.begin_step()
        with g.as_default():
          logits = self.model_helper.forward_eval(to_device_images(images, g))
          loss, metrics = self.model_helper.calc_loss(labels.to(self.device), logits, self.vars_prnd['trainable'])
        names = list(metrics.keys())
        rows.append([float(loss)] + [float(v) for v in metrics.values()])
    means = np.mean(np.array(rows, dtype=np.float64), axis=0)
    return means[0], {name: means[idx + 1] for idx, name in enumerate(names)}
