"""Running mean / standard deviation (reference rl_agents/ddpg/running_mean_std.py:25-69).

The reference agent hard-codes `normalize_state = normalize_return = False` (agent.py:262-263), so this is only
used when a caller switches the normalisation on explicitly."""
import numpy as np

from pocketflow_amd.flags import FLAGS, flags

flags.DEFINE_float('ddpg_rms_eps', 1e-4, 'DDPG: running standard deviation\'s epsilon')


class RunningMeanStd(object):
  def __init__(self, sess, nb_dims):
    self.x_sum = np.zeros(nb_dims, np.float32)
    self.x_sum_sq = np.zeros(nb_dims, np.float32)
    self.x_cnt = np.float32(0)

  @property
  def mean(self):
    return np.zeros_like(self.x_sum) if self.x_cnt < 0.5 else self.x_sum / self.x_cnt

  @property
  def std(self):
    if self.x_cnt < 0.5:
      return np.ones_like(self.x_sum)
    return np.sqrt(np.maximum(self.x_sum_sq / self.x_cnt - np.square(self.mean), np.float32(FLAGS.ddpg_rms_eps)))

  def updt(self, x_new):
    x_new = np.asarray(x_new, np.float32)
    self.x_sum += x_new.sum(axis=0)
    self.x_sum_sq += np.square(x_new).sum(axis=0)
    self.x_cnt += np.float32(x_new.shape[0])
