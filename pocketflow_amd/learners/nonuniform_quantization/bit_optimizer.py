"""Bit allocation for the non-uniform quantisation learner (reference nonuniform bit_optimizer.py:53+).
Non-RL branch only: every quantised matmul gets `nuql_weight_bits`, activations `nuql_activation_bits`."""
from pocketflow_amd.flags import FLAGS, flags

flags.DEFINE_integer('nuql_equivalent_bits', 4, 'equivalent compression bits for non-rl quantization')
flags.DEFINE_integer('nuql_nb_rlouts', 200, 'total number of rlouts for rl training')
flags.DEFINE_integer('nuql_w_bit_min', 2, 'minimum number of bits for weights')
flags.DEFINE_integer('nuql_w_bit_max', 8, 'maximum number of bits for weights')
flags.DEFINE_integer('nuql_tune_layerwise_steps', 100, 'fine tuning steps for each layer')
flags.DEFINE_integer('nuql_tune_global_steps', 2101, 'fine tuning steps for each layer')
flags.DEFINE_string('nuql_tune_save_path', './rl_tune_models/model.ckpt', 'dir to save tuned models during rl trianing')
flags.DEFINE_integer('nuql_tune_disp_steps', 300, 'interval steps to show tuning details')
flags.DEFINE_boolean('nuql_enbl_random_layers', True, 'enable random permutation of layers for the rl agent')
flags.DEFINE_boolean('nuql_enbl_rl_agent', False, 'enable rl agent for non-uniform quantization')
flags.DEFINE_boolean('nuql_enbl_rl_global_tune', True, 'Tune the weights of all layers in the rl training')
flags.DEFINE_boolean('nuql_enbl_rl_layerwise_tune', False, 'Tune the weights of each layers in the rl training')


class BitOptimizer(object):
  def __init__(self, dataset_name, weights, statistics, *unused):
    self.dataset_name = dataset_name
    self.weights = weights
    self.statistics = statistics

  def run(self):
    if FLAGS.nuql_enbl_rl_agent:
      raise NotImplementedError('the DDPG bit allocator is outside the MI355X hot path (SURVEY 8f row 2)')
    return ([FLAGS.nuql_weight_bits] * self.statistics['nb_matmuls'],
            [FLAGS.nuql_activation_bits] * self.statistics['nb_activations'])
