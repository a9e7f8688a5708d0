"""Exploration-noise schedules of the DDPG agent (reference rl_agents/ddpg/noise.py:24-86).

`stdev_curr` is the standard deviation of the parameter (or action) noise of the next roll-out.
  tdecy: multiplied by (std_finl / std_init)^(1 / nb_rlouts) once per roll-out after exploration has ended;
  adapt: divided / multiplied by ddpg_noise_adpt_rat so that the mean |clean - perturbed action| tracks ddpg_noise_dst_finl."""
from pocketflow_amd.flags import FLAGS, flags

flags.DEFINE_string('ddpg_noise_type', 'param', 'DDPG: noise type (\'action\' OR \'param\')')
flags.DEFINE_string('ddpg_noise_prtl', 'tdecy', 'DDPG: noise adjustment protocol (\'adapt\' OR \'tdecy\')')
flags.DEFINE_float('ddpg_noise_std_init', 1e+0, 'DDPG: parameter / action noise\'s initial stdev.')
flags.DEFINE_float('ddpg_noise_dst_finl', 1e-2, 'DDPG: action noise\'s final distance')
flags.DEFINE_float('ddpg_noise_adpt_rat', 1.03, 'DDPG: parameter noise\'s adaption rate')
flags.DEFINE_float('ddpg_noise_std_finl', 1e-5, 'DDPG: parameter / action noise\'s final stdev.')


class _NoiseSpec(object):
  def __init__(self):
    self.reset()

  def reset(self):
    self.stdev_curr = FLAGS.ddpg_noise_std_init


class AdaptiveNoiseSpec(_NoiseSpec):
  def adapt(self, dst_curr):
    too_far = dst_curr > FLAGS.ddpg_noise_dst_finl
    self.stdev_curr = self.stdev_curr / FLAGS.ddpg_noise_adpt_rat if too_far else self.stdev_curr * FLAGS.ddpg_noise_adpt_rat


class TimeDecayNoiseSpec(_NoiseSpec):
  def __init__(self, nb_rlouts):
    super(TimeDecayNoiseSpec, self).__init__()
    self.decy_rat = (FLAGS.ddpg_noise_std_finl / FLAGS.ddpg_noise_std_init) ** (1.0 / nb_rlouts)

  def adapt(self):
    self.stdev_curr *= self.decy_rat
