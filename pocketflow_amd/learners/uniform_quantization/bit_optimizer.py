"""Bit allocation for the uniform-quantisation learner (reference bit_optimizer.py:50-135).

Only the non-RL branch is on the hot path: every quantised matmul gets `uql_weight_bits`, every
activation `uql_activation_bits` (:128-135).  The DDPG search (:137-366) is a SURVEY 8f "next" row.
"""
from pocketflow_amd.flags import FLAGS, flags

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


class BitOptimizer(object):
  def __init__(self, dataset_name, weights, statistics, *unused):
    self.dataset_name = dataset_name
    self.weights = weights
    self.statistics = statistics

  def run(self):
    """Return (w_bit_list, a_bit_list)."""
    if FLAGS.uql_enbl_rl_agent:
      raise NotImplementedError('the DDPG bit allocator is outside the MI355X hot path (SURVEY 8f row 2); '
                                'run with --nouql_enbl_rl_agent')
    optimal_w_bit_list = [FLAGS.uql_weight_bits] * self.statistics['nb_matmuls']
    optimal_a_bit_list = [FLAGS.uql_activation_bits] * self.statistics['nb_activations']
    return optimal_w_bit_list, optimal_a_bit_list
