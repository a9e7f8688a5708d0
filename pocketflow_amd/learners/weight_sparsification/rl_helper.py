"""State / action bookkeeping of the pruning-ratio search (reference learners/weight_sparsification/rl_helper.py:24-161).

State of maskable tensor i, every column divided by its maximum over i (the "kept so far" column by the maximum of the
"still to come" column):
  one-hot(i) | kernel shape (4) | n_i | sum_{j<i} n_j (1 - r_j)  [dynamic] | sum_{j>i} n_j
Action a in [0, 1] -> pruning ratio: piecewise linear through (0, r_lo), (0.5, ws_prune_ratio), (1, r_hi), clamped to
[r_lo, r_hi] with r_lo = max(0, 1 - 3 (1 - r)), r_hi = 1 - (1 - r) / 3; under the single-objective reward r_lo is
raised so that the overall target stays reachable with every later tensor pruned at its r_hi."""
import numpy as np

from pocketflow_amd.flags import FLAGS
from pocketflow_amd.learners.uniform_quantization.rl_helper import kernel_shape4


class RLHelper(object):
  def __init__(self, sess, maskable_vars, skip_head_n_tail):
    n = len(maskable_vars)
    shapes = np.stack([kernel_shape4(v) for v in maskable_vars])
    self.nb_params_full = shapes.prod(axis=1)
    self.prune_ratios = np.zeros(n)
    self.s_dims = n + 4 + 3
    behind = self.nb_params_full[::-1].cumsum()[::-1] - self.nb_params_full
    self.states = np.hstack([np.eye(n), shapes, self.nb_params_full[:, None], np.zeros((n, 1)), behind[:, None]])
    self.state_normalizer = self.states.max(axis=0)
    self.state_normalizer[-2] = self.state_normalizer[-1]

    keep = 1.0 - FLAGS.ws_prune_ratio
    self.prune_ratios_min = np.full(n, max(0.0, 1.0 - keep * 3.0))
    self.prune_ratios_max = np.full(n, 1.0 - keep / 3.0)
    if skip_head_n_tail:                                       # first & last layer stay dense (CIFAR-10 nets)
      for bounds in (self.prune_ratios_min, self.prune_ratios_max):
        bounds[[0, -1]] = 0.0

  def calc_state(self, idx):
    row = self.states[idx].copy()
    row[-2] = np.sum(self.nb_params_full[:idx] * (1.0 - self.prune_ratios[:idx]))
    return (row / self.state_normalizer)[None, :]

  def calc_overall_prune_ratio(self):
    return np.sum(self.nb_params_full * self.prune_ratios) / np.sum(self.nb_params_full)

  def calc_reward(self, accuracy):
    if FLAGS.ws_reward_type == 'single-obj':
      return accuracy
    if FLAGS.ws_reward_type == 'multi-obj':
      return accuracy * np.log(1.0 + self.calc_overall_prune_ratio())
    raise ValueError('unrecognized reward type: ' + FLAGS.ws_reward_type)

  def _bounds(self, idx):
    lo, hi = self.prune_ratios_min[idx], self.prune_ratios_max[idx]
    if FLAGS.ws_reward_type == 'single-obj':
      n = self.nb_params_full
      best_case_elsewhere = np.sum(n[:idx] * self.prune_ratios[:idx]) + np.sum(n[idx + 1:] * self.prune_ratios_max[idx + 1:])
      required = (np.sum(n) * FLAGS.ws_prune_ratio - best_case_elsewhere) / n[idx]
      assert required < hi + 1e-4, 'cannot reach the required pruning ratio: %f vs. %f' % (required, hi)
      lo = max(lo, required)
    return lo, hi

  def cvt_action_to_prune_ratio(self, idx, action):
    lo, hi = self._bounds(idx)
    mid = FLAGS.ws_prune_ratio
    if action > 0.5:
      ratio = hi - (1.0 - action) / 0.5 * (hi - mid)
    else:
      ratio = lo + (action - 0.0) / 0.5 * (mid - lo)
    self.prune_ratios[idx] = max(lo, min(hi, ratio))
    return self.prune_ratios[idx]
