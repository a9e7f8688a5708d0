"""Bit allocation for the non-uniform quantisation learner (reference learners/nonuniform_quantization/bit_optimizer.py:53-371).

Same DDPG search as the uniform learner's (see learners/uniform_quantization/bit_optimizer.py) with the differences of
the reference's non-uniform copy:
  * start: `ops['non_cluster_init']` only, no broadcast (:145);
  * per roll-out: [restore the pre-trained weights if enbl_warm_start] -> `ops['cluster_init']` WITH the roll-out's
    bit widths (codebook sizes follow them) -> broadcast (:227-238);
  * the fine-tune runs `ops['rl_fintune']` -- plain SGD when the codebooks are optimised (`nuql_opt_mode` cluster /
    both; Adam slots cannot follow codebooks that change size), the regular Adam train op otherwise (nuq
    learner.py:262-285);
  * the best reward starts at -1 (:155)."""
from __future__ import annotations

import os

from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.learners.nonuniform_quantization.rl_helper import RLHelper
from pocketflow_amd.learners.uniform_quantization.bit_optimizer import BitOptimizer as _UqBitOptimizer
from pocketflow_amd.utils import checkpoint

flags.DEFINE_integer('nuql_equivalent_bits', 4, 'equivalent compression bits for non-rl quantization')
flags.DEFINE_integer('nuql_nb_rlouts', 200, 'total number of rlouts for rl training')
flags.DEFINE_integer('nuql_w_bit_min', 2, 'minimum number of bits for weights')
flags.DEFINE_integer('nuql_w_bit_max', 8, 'maximum number of bits for weights')
flags.DEFINE_integer('nuql_tune_layerwise_steps', 100, 'fine tuning steps for each layer')
flags.DEFINE_integer('nuql_tune_global_steps', 2101, 'fine tuning steps for all layers')
flags.DEFINE_string('nuql_tune_save_path', './rl_tune_models/model.ckpt', 'dir to save tuned models during rl trianing')
flags.DEFINE_integer('nuql_tune_disp_steps', 300, 'interval steps to show tuning details')
flags.DEFINE_boolean('nuql_enbl_random_layers', True, 'enable random permutation of layers for the rl agent')
flags.DEFINE_boolean('nuql_enbl_rl_agent', False, 'enable rl agent for non-uniform quantization')
flags.DEFINE_boolean('nuql_enbl_rl_global_tune', True, 'Tune the weights of all layers in the rl training')
flags.DEFINE_boolean('nuql_enbl_rl_layerwise_tune', False, 'Tune the weights of each layers in the rl training')


class BitOptimizer(_UqBitOptimizer):
  PREFIX = 'nuql'
  HELPER = RLHelper

  def _begin_search(self):
    self.ops['non_cluster_init']()

  def _optimal_reward_init(self):
    return -1

  def _restore_for_finetune(self, layer_bits):
    if FLAGS.enbl_warm_start:
      self.ops['restore'](checkpoint.latest_checkpoint(os.path.dirname(FLAGS.save_path)))
    self.ops['cluster_init'](layer_bits)
    if FLAGS.enbl_multi_gpu and self.ops.get('bcast'):
      self.ops['bcast']()

  def _train_op(self):
    return self.ops['rl_fintune']
