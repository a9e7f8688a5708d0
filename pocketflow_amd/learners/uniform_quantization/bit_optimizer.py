"""Bit allocation for the uniform-quantisation learner (reference bit_optimizer.py:50-366).

Without `--uql_enbl_rl_agent` every quantised matmul gets `uql_weight_bits`, every activation
`uql_activation_bits` (:128-135).  With it a DDPG agent (pocketflow_amd/rl_agents/ddpg) searches per-layer weight
bit widths under the budget `sum_i n_i * uql_equivalent_bits` (:137-195).  One roll-out:

  rank 0: for each layer (random order): state -> noisy actor -> RLHelper.calc_w -> bits      (:255-279)
  all   : bits broadcast; restore the pre-trained weights; `uql_tune_global_steps / world` quantisation-aware
          Adam steps WITH those bits (= the hot path: the fused quantisers take per-layer bit widths from the
          segment table, nothing is rebuilt); reset the fine-tune step                        (:197-254)
  rank 0: reward = eval top-1 (CIFAR-10) / top-5 (ILSVRC-12) over nb_smpls_eval // batch_size_eval batches with
          activations at 32 bits; record the transitions, `nb_matmuls` agent updates           (:215-231, 281-310)

Differences from the reference, all on the control plane: the bit list travels through `mpi_comm.bcast` (RCCL /
gloo object broadcast) instead of `./arranged_layer_bits.txt`; the learner hands over callables instead of TF ops
and sessions (`ops['train'](w_bits, a_bits)`, `ops['eval'](w_bits, a_bits)`, `ops['restore'](path)`,
`ops['layerwise_tune'](n, w_bits, a_bits)`, ...); after the rank-0-only layer-wise fine-tune
(`uql_enbl_rl_layerwise_tune`, off by default, "working not very well" in the reference) the variables are broadcast
so that the ranks stay consistent (the reference leaves them diverged, its TODO at :206).
"""
from __future__ import annotations

import logging
import os
from timeit import default_timer as timer

import numpy as np

from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.learners.uniform_quantization.rl_helper import RLHelper
from pocketflow_amd.rl_agents.ddpg.agent import Agent as DdpgAgent
from pocketflow_amd.utils import checkpoint
from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw

flags.DEFINE_integer('uql_equivalent_bits', 4, 'equivalent compression bits for non-rl quantization')
flags.DEFINE_integer('uql_nb_rlouts', 200, 'total number of rlouts for rl training')
flags.DEFINE_integer('uql_w_bit_min', 2, 'minimum number of bits for weights')
flags.DEFINE_integer('uql_w_bit_max', 8, 'maximum number of bits for weights')
flags.DEFINE_integer('uql_tune_layerwise_steps', 100, 'fine tuning steps for each layer')
flags.DEFINE_integer('uql_tune_global_steps', 2000, 'fine tuning steps for each layer')
flags.DEFINE_string('uql_tune_save_path', './rl_tune_models/model.ckpt', 'dir to save tuned models during rl trianing')
flags.DEFINE_integer('uql_tune_disp_steps', 300, 'interval steps to show tuning details')
flags.DEFINE_boolean('uql_enbl_random_layers', True, 'enable random permutation of layers for the rl agent')
flags.DEFINE_boolean('uql_enbl_rl_agent', False, 'enable rl agent for uniform quantization')
flags.DEFINE_boolean('uql_enbl_rl_global_tune', True, 'Tune the weights of all layers in the rl training')
flags.DEFINE_boolean('uql_enbl_rl_layerwise_tune', False, 'Tune the weights of each layers in the rl training')

log = logging.getLogger('pocketflow_amd')


class BitOptimizer(object):
  # pylint: disable=too-many-instance-attributes
  """Currently only weight bits are inferred via RL; activations stay at 32 bits during the search."""
  PREFIX = 'uql'
  HELPER = RLHelper

  def __init__(self, dataset_name, weights, statistics, bit_placeholders=None, ops=None, layerwise_tune_list=(None, None),
               sess_train=None, sess_eval=None, saver_train=None, saver_eval=None, barrier_fn=None, mpi_comm=None):
    self.dataset_name = dataset_name
    self.weights = weights
    self.statistics = statistics
    self.bit_placeholders = bit_placeholders
    self.ops = ops or {}
    self.auto_barrier = barrier_fn or (lambda: None)
    self.mpi_comm = mpi_comm
    self.total_num_weights = sum(self.statistics['num_weights'])
    self.total_bits = self.total_num_weights * self._flag('equivalent_bits')
    self.mgw_size = int(mgw.size()) if FLAGS.enbl_multi_gpu else 1
    self.tune_global_steps = int(self._flag('tune_global_steps') / self.mgw_size)
    self.tune_global_disp_steps = max(int(self._flag('tune_disp_steps') / self.mgw_size), 1)
    self.agent = None
    if self._flag('enbl_rl_agent'):
      self.__build_agent()

  @classmethod
  def _flag(cls, name):
    return getattr(FLAGS, '%s_%s' % (cls.PREFIX, name))

  def __build_agent(self):
    self.w_rl_helper = self.HELPER(None, self.total_bits, self.statistics['num_weights'], self.weights,
                                   random_layers=self._flag('enbl_random_layers'))
    self.s_dims = self.w_rl_helper.s_dims
    self.a_dims = 1
    buff_size = len(self.weights) * int(self._flag('nb_rlouts') // 4)
    if buff_size < 1:
      raise ValueError('%s_nb_rlouts must be at least 4 (the replay buffer holds nb_matmuls * nb_rlouts // 4 rows)' % self.PREFIX)
    self.agent = DdpgAgent(None, self.s_dims, self.a_dims,
                           self._flag('nb_rlouts'), buff_size, a_min=0., a_max=self._flag('w_bit_max') - self._flag('w_bit_min'))

  def run(self):
    """Return (w_bit_list, a_bit_list), either searched by the RL agent or constant."""
    if self._flag('enbl_rl_agent'):
      return self._calc_optimal_bits()
    optimal_w_bits = [self._flag('weight_bits')] * self.statistics['nb_matmuls']
    optimal_a_bits = [self._flag('activation_bits')] * self.statistics['nb_activations']
    return optimal_w_bits, optimal_a_bits

  # -- the search ---------------------------------------------------------------------------------------------------------
  def _begin_search(self):
    self.ops['init']()
    if FLAGS.enbl_multi_gpu and self.ops.get('bcast'):
      self.ops['bcast']()

  def _optimal_reward_init(self):
    return -np.inf

  def _calc_optimal_bits(self):
    self._begin_search()
    fp_a_bit_list = [32] * self.statistics['nb_activations']
    primary = self.__is_primary_worker()
    if primary:
      self.agent.init()
      self.reward_list = []
    optimal_reward, optimal_arranged_w_bit_list = self._optimal_reward_init(), None

    for idx_rlout in range(self._flag('nb_rlouts')):
      states_n_actions, arranged_layer_bits = None, None
      if primary:
        log.info('starting %d-th roll-out:' % idx_rlout)
        states_n_actions, arranged_layer_bits = self.__calc_rollout_actions(idx_rlout)
      arranged_layer_bits = self.__sync(arranged_layer_bits)
      reward = self._calc_rollout_reward(arranged_layer_bits, fp_a_bit_list)
      self.auto_barrier()
      if primary:
        self.reward_list.append(reward[0][0])
        self.agent.finalize_rlout(reward)
        self.__record_rollout_transitions(states_n_actions, reward)
        self.__train_rl_agent(idx_rlout)
        if optimal_reward < reward:
          optimal_reward = reward
          optimal_arranged_w_bit_list = arranged_layer_bits
      self.auto_barrier()

    if primary:
      log.info("Finished RL training")
      log.info("Optimal reward: {0}, Optimal w_bit_list: {1}".format(optimal_reward, optimal_arranged_w_bit_list))
    optimal_arranged_w_bit_list = self.__sync(optimal_arranged_w_bit_list)
    return optimal_arranged_w_bit_list, fp_a_bit_list

  def __sync(self, bit_list):
    """`__sync_list_write` + barrier + `__sync_list_read` of the reference (:353-366): rank 0's list, rounded."""
    if FLAGS.enbl_multi_gpu and self.mpi_comm is not None:
      bit_list = self.mpi_comm.bcast(bit_list, root=0)
    return [round(float(bit)) for bit in bit_list]

  def _restore_for_finetune(self, layer_bits):
    save_path = checkpoint.latest_checkpoint(os.path.dirname(FLAGS.save_path))
    self.ops['restore'](save_path)
    if FLAGS.enbl_multi_gpu and self.ops.get('bcast'):
      self.ops['bcast']()

  def _calc_rollout_reward(self, layer_bits, fp_a_bit_list):
    if self._flag('enbl_rl_global_tune') or self._flag('enbl_rl_layerwise_tune'):
      self._restore_for_finetune(layer_bits)
    if self._flag('enbl_rl_layerwise_tune'):
      if self.__is_primary_worker():
        self.__layerwise_finetune(layer_bits, fp_a_bit_list)
      self.auto_barrier()
      if FLAGS.enbl_multi_gpu and self.ops.get('bcast'):
        self.ops['bcast']()
    self.auto_barrier()
    if self._flag('enbl_rl_global_tune'):
      self._global_finetune(layer_bits, fp_a_bit_list)
    if not self.__is_primary_worker():
      return None
    self.ops['save'](self._flag('tune_save_path'))
    __, acc_top1, acc_top5 = self.__calc_loss_n_accuracy(layer_bits, fp_a_bit_list)
    if self.dataset_name == 'cifar_10':
      reward = self.w_rl_helper.calc_reward(acc_top1)
    elif self.dataset_name == 'ilsvrc_12':
      reward = self.w_rl_helper.calc_reward(acc_top5)
    else:
      raise ValueError("Unknown dataset name")
    log.info('acc_top1 = %.4f | acc_top5 = %.4f | reward = %.4f' % (acc_top1, acc_top5, reward[0][0]))
    return reward

  def _train_op(self):
    return self.ops['train']

  def __layerwise_finetune(self, layer_bits, fp_a_bit_list):
    """Rank 0 only: *_tune_layerwise_steps steps of every layer's tune op (:233-243)."""
    for n in range(self.statistics['nb_matmuls']):
      for t_step in range(self._flag('tune_layerwise_steps')):
        diff = self.ops['layerwise_tune'](n, layer_bits, fp_a_bit_list)
        if (t_step + 1) % 20 == 0:
          log.info("Layerwise Tuning: {}, Step: {}, Bit: {}, Layer diff norm: {}".format(n, t_step + 1, layer_bits[n], diff))
    log.info("Layerwise finetuning done")

  def _global_finetune(self, layer_bits, fp_a_bit_list):
    time_prev = timer()
    for t_step in range(self.tune_global_steps):
      log_rslt = self._train_op()(layer_bits, fp_a_bit_list)
      if (t_step + 1) % self.tune_global_disp_steps == 0:
        time_prev = self.__monitor_progress(t_step, log_rslt, time_prev)
    self.ops['reset_ft_step']()      # so that the learning-rate schedule restarts

  def __calc_rollout_actions(self, idx_rlout):
    """Rank 0 only: one bit width per layer from the noisy actor, made feasible by the helper."""
    self.agent.init_rlout()
    self.w_rl_helper.reset()
    nb = self.statistics['nb_matmuls']
    states_n_actions = [(None, None)] * nb           # indexed by layer, whatever the visiting order
    arranged = [-1] * nb
    for idx in self.w_rl_helper.layer_idxs:
      state = self.w_rl_helper.calc_state(idx)
      action = self.w_rl_helper.calc_w(self.agent.actions_noisy(state), idx)
      assert np.shape(action) == (1, 1), '"action" must be in shape (1,1)'
      assert 1 <= action[0][0] <= 32, 'the quantization bits must be in [1, 32]'
      states_n_actions[idx] = (state, action)
      arranged[idx] = float(action[0][0])
    assert -1 not in arranged, "Some layers are not assigned with proper bits"
    log.info('Un-allocated bit percentage: %.3f' % self.__check_bits(arranged))
    log.info('#_rlout: {0}, layer_bits: {1}'.format(idx_rlout, arranged))
    return states_n_actions, arranged

  def __train_rl_agent(self, idx_rlout):
    for _ in range(self.statistics['nb_matmuls']):
      actor_loss, critic_loss, param_noise_std = self.agent.train()
    log.info('roll-out #%d: a-loss = %.2e | c-loss = %.2e | noise std. = %.2e'
             % (idx_rlout, actor_loss, critic_loss, param_noise_std))

  def __calc_loss_n_accuracy(self, w_bits, a_bits):
    """Mean loss / top-1 / top-5 over nb_smpls_eval // batch_size_eval evaluation batches (:300-310)."""
    nb_iters = FLAGS.nb_eval_batches_override or FLAGS.nb_smpls_eval // FLAGS.batch_size_eval
    rows = np.array([self.ops['eval'](w_bits, a_bits) for _ in range(nb_iters)], dtype=np.float64)
    return rows[:, 0].mean(), rows[:, 1].mean(), rows[:, 2].mean()

  def __record_rollout_transitions(self, states_n_actions, reward):
    nb = self.statistics['nb_matmuls']
    for n, (state, action) in enumerate(states_n_actions):
      last = n == nb - 1
      terminal = np.ones((1, 1)) if last else np.zeros((1, 1))
      state_next = np.zeros((1, self.s_dims)) if last else states_n_actions[n + 1][0]
      self.agent.record(state, action, reward, terminal, state_next)

  def __check_bits(self, bit_list):
    used_bits = sum(v * p for v, p in zip(bit_list, self.statistics['num_weights']))
    if self.total_bits < used_bits:
      raise ValueError("The average bit is out of constraint")
    return (self.total_bits - used_bits) / self.total_bits

  def __monitor_progress(self, idx_iter, log_rslt, time_prev):
    if not self.__is_primary_worker():
      return None
    speed = FLAGS.batch_size * self.tune_global_disp_steps / (timer() - time_prev) * self.mgw_size
    names = ['lr', 'dst_loss', 'model_loss', 'loss', 'acc_top1', 'acc_top5'] if FLAGS.enbl_dst else \
        ['lr', 'model_loss', 'loss', 'acc_top1', 'acc_top5']
    log.info('iter #%d: %s | speed = %.2f pics / sec', idx_iter + 1,
             ' | '.join('%s = %e' % (k, float(v)) for k, v in zip(names, log_rslt)), speed)
    return timer()

  @classmethod
  def __is_primary_worker(cls):
    return not FLAGS.enbl_multi_gpu or mgw.rank() == 0
