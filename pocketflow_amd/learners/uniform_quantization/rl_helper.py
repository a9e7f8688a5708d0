"""State / action bookkeeping of the bit-allocation search (reference learners/uniform_quantization/rl_helper.py:26-122;
the non-uniform copy at learners/nonuniform_quantization/rl_helper.py differs only in the flag prefix).

State of layer i (a row of `self.states`, built once):
  one-hot(i) | kernel shape (4; fully-connected kernels get two leading ones) | n_i / max_j n_j | sum_{j>i} n_j / sum_j n_j
Action -> bits: `round(a) + w_bit_min`, capped so that every layer still to come can get at least w_bit_min bits
inside the budget `sum_j n_j * equivalent_bits`; the last layer visited in a roll-out takes the floor of what is
left; everything is capped at w_bit_max."""
import random

import numpy as np

from pocketflow_amd.flags import FLAGS


def kernel_shape4(var):
  """Reference-layout kernel shape as 4 numbers: [kh, kw, cin, cout], or [1, 1, cin, cout] for a dense kernel."""
  dims = [float(d) for d in (getattr(var, 'ref_shape', None) or var.shape)]
  assert len(dims) in (2, 4), "Unknown weight shape. Must be a 2 (fc) or 4 (conv) dimensional."
  return np.array([1.0] * (4 - len(dims)) + dims)


class RLHelper(object):
  FLAG_PREFIX = 'uql'

  def __init__(self, sess, total_bits, num_weights, vars_list, random_layers=False):
    """`sess` is unused (reference signature); `vars_list` holds graph Variables (`.ref_shape`) or arrays."""
    n = len(num_weights)
    counts = np.asarray(num_weights, dtype=np.float64)
    self.nb_vars, self.num_weights = n, num_weights
    self.total_num_weights = sum(num_weights)
    self.total_bits = total_bits
    self.random_layers = random_layers
    self.layer_idxs = list(range(n))
    self.shuffle = random.shuffle                    # the reference draws from the global `random` generator
    self.var_shapes = [kernel_shape4(v) for v in vars_list]
    self.s_dims = n + 6
    after = counts[::-1].cumsum()[::-1] - counts                  # parameters of the layers behind i
    self.states = np.hstack([np.eye(n), np.stack(self.var_shapes), (counts / counts.max())[:, None],
                             (after / self.total_num_weights)[:, None]])
    self.reset(shuffle=False)

  def _flag(self, name):
    return getattr(FLAGS, '%s_%s' % (self.FLAG_PREFIX, name))

  def calc_state(self, idx):
    return self.states[idx:idx + 1].copy()

  def calc_reward(self, accuracy):
    return np.full((1, 1), accuracy, dtype=np.float64)

  def reset(self, shuffle=True):
    """Start of a roll-out: empty budget counters; with `random_layers` the layers are visited in a new order."""
    self.w_bits_used = 0
    self.quantized_layers = 0
    self.num_weights_to_quantize = self.total_num_weights
    if shuffle and self.random_layers:
      self.shuffle(self.layer_idxs)

  def calc_w(self, action, idx):
    """Actor output (1, 1) -> feasible number of bits (1, 1) for layer `idx`; updates the budget counters."""
    lo, hi, n_i = self._flag('w_bit_min'), self._flag('w_bit_max'), self.num_weights[idx]
    spare = self.total_bits - self.w_bits_used - self.num_weights_to_quantize * lo   # bits beyond the guaranteed minimum
    assert spare >= 0, "Not enough budget for layer {}".format(idx)
    if self.quantized_layers == self.nb_vars - 1:
      bits = np.full((1, 1), np.floor((self.total_bits - self.w_bits_used) / n_i))
    else:
      bits = np.minimum(np.round(action) + lo, lo + np.floor(spare * 1.0 / n_i))
    bits = np.minimum(bits, hi)
    self.w_bits_used += bits[0][0] * n_i
    self.num_weights_to_quantize -= n_i
    self.quantized_layers += 1
    return bits
