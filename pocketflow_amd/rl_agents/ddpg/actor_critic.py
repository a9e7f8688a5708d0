"""Actor / critic networks of the DDPG agent (reference rl_agents/ddpg/actor_critic.py:24-154).

Both are 3-4 layer MLPs of width 64 evaluated on ONE state row per decision and on 64-row mini-batches
per update: a few hundred kFLOP, bound by launch latency on any accelerator.  They are therefore plain
float32 torch tensors on the host by default (`--ddpg_device cpu`); the roll-outs they steer are what
runs on the MI355X.

Variables carry the reference's names (`<scope>/dense[_k]/{kernel,bias}`, `<scope>/LayerNorm[_k]/{beta,gamma}`)
in creation order, which is what the parameter-noise rule ("perturb every trainable variable whose name does
not contain LayerNorm", :75-78) and the checkpoints key on.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from pocketflow_amd.flags import FLAGS, flags

flags.DEFINE_integer('ddpg_actor_depth', 2, 'DDPG: actor network\'s depth')
flags.DEFINE_integer('ddpg_actor_width', 64, 'DDPG: actor network\'s width')
flags.DEFINE_integer('ddpg_critic_depth', 2, 'DDPG: critic network\'s depth')
flags.DEFINE_integer('ddpg_critic_width', 64, 'DDPG: critic network\'s width')
flags.DEFINE_string('ddpg_device', 'cpu', 'DDPG: device of the actor / critic tensors (cpu | cuda)')

ENBL_LAYER_NORM = True
LAYER_NORM_EPS = 1e-12        # tf.contrib.layers.layer_norm's variance_epsilon


class Model(object):
  """A bag of named float32 tensors + a builder that hands them out in creation order."""

  def __init__(self, scope, rng=None, device=None):
    self.scope = scope
    self.rng = rng if rng is not None else np.random
    self.device = torch.device(device or FLAGS.ddpg_device)
    self.params: 'OrderedDict[str, torch.Tensor]' = OrderedDict()
    self._counts = {}

  # -- the reference's three views ---------------------------------------------------------------------
  @property
  def vars(self):
    return list(self.params.values())

  @property
  def var_names(self):
    return list(self.params.keys())

  @property
  def trainable_vars(self):
    return list(self.params.values())

  @property
  def perturbable_vars(self):
    return [v for k, v in self.params.items() if 'LayerNorm' not in k]

  # -- variable creation (first call) / lookup (later calls, = reuse) ----------------------------------------
  def _begin(self):
    self._counts = {}

  def _layer(self, base):
    k = self._counts.get(base, 0)
    self._counts[base] = k + 1
    return '%s/%s' % (self.scope, base if k == 0 else '%s_%d' % (base, k))

  def _get(self, name, shape, init):
    if name not in self.params:
      self.params[name] = torch.tensor(init(shape), dtype=torch.float32, device=self.device, requires_grad=True)
    p = self.params[name]
    assert tuple(p.shape) == tuple(shape), '%s: %s vs %s' % (name, tuple(p.shape), tuple(shape))
    return p

  def _glorot(self, shape):
    limit = np.sqrt(6.0 / (shape[0] + shape[1]))       # tf.layers.dense's default initialiser
    return self.rng.uniform(-limit, limit, shape).astype(np.float32)

  def dense(self, x, units):
    layer = self._layer('dense')
    w = self._get(layer + '/kernel', (x.shape[1], units), self._glorot)
    b = self._get(layer + '/bias', (units,), lambda s: np.zeros(s, np.float32))
    return x @ w + b

  def layer_norm(self, x):
    layer = self._layer('LayerNorm')
    beta = self._get(layer + '/beta', (x.shape[1],), lambda s: np.zeros(s, np.float32))
    gamma = self._get(layer + '/gamma', (x.shape[1],), lambda s: np.ones(s, np.float32))
    mean = x.mean(dim=1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + LAYER_NORM_EPS) * gamma + beta

  def dense_block(self, x, units):
    x = self.dense(x, units)
    if ENBL_LAYER_NORM:
      x = self.layer_norm(x)
    return torch.relu(x)

  def reinitialize(self):
    """tf.variables_initializer(self.vars): fresh draws for the kernels, constants elsewhere."""
    with torch.no_grad():
      for name, p in self.params.items():
        if name.endswith('/kernel'):
          p.copy_(torch.from_numpy(self._glorot(tuple(p.shape))))
        elif name.endswith('/gamma'):
          p.fill_(1.0)
        else:
          p.zero_()

  def export_numpy(self):
    return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in self.params.items())

  def load_numpy(self, values):
    with torch.no_grad():
      for k, p in self.params.items():
        p.copy_(torch.from_numpy(np.asarray(values[k], np.float32)))


class Actor(Model):
  """states -> depth x [Dense + LayerNorm + ReLU] -> Dense(a_dims) -> sigmoid scaled to [a_min, a_max]."""

  def __init__(self, a_dims, a_min, a_max, scope='actor', rng=None, device=None):
    super(Actor, self).__init__(scope, rng, device)
    self.a_dims, self.a_min, self.a_max = a_dims, a_min, a_max

  def __call__(self, states, reuse=False):
    self._begin()
    x = states
    for __ in range(FLAGS.ddpg_actor_depth):
      x = self.dense_block(x, FLAGS.ddpg_actor_width)
    x = self.dense(x, self.a_dims)
    return torch.sigmoid(x) * (self.a_max - self.a_min) + self.a_min


class Critic(Model):
  """[Dense + LN + ReLU](states) (+) actions -> depth x [Dense + LN + ReLU] -> Dense(1)."""

  def __init__(self, scope='critic', rng=None, device=None):
    super(Critic, self).__init__(scope, rng, device)

  def __call__(self, states, actions, reuse=False):
    self._begin()
    x = self.dense_block(states, FLAGS.ddpg_critic_width)
    x = torch.cat([x, actions], dim=1)
    for __ in range(FLAGS.ddpg_critic_depth):
      x = self.dense_block(x, FLAGS.ddpg_critic_width)
    return self.dense(x, 1)
