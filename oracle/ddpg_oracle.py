"""CPU oracle for the DDPG hyper-parameter search agent  --  TEST INFRASTRUCTURE ONLY.

Plain-NumPy restatement (float32, hand-written backward passes: no autograd) of the actor / critic networks and
of one agent update of the reference (paths relative to /root/reference):
  rl_agents/ddpg/actor_critic.py:30-154   dense_block, Actor.__call__, Critic.__call__
  rl_agents/ddpg/agent.py:71-117, 216-247, 280-300, 372-408   target update, parameter noise, train(), losses

PARITY.  The forward passes are pinned against the reference's own `Actor` / `Critic` classes executed over
oracle/tf_stub.py, and the UPDATE STEP against the reference's own `Agent` class (__build, init, record, finalize_rlout,
train) executed over the deferred-execution stand-in oracle/tf_graph_stub.py (placeholders, optimizer.minimize,
tf.assign, sess.run) -- tests/golden/make_reference_rl_golden.py -> tests/golden/reference_rl.npz; checked in
tests/test_rl_golden.py: losses and every variable of the main and target networks after each of three updates.
What stays a restatement is TensorFlow's own kernels (tf.layers.dense, layer_norm, AdamOptimizer arithmetic), as for
every other fixture of this repository: TF never ran here.

Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
LN_EPS = F32(1e-12)            # tf.contrib.layers.layer_norm -> tf.nn.batch_normalization(variance_epsilon=1e-12)


# -- layers with explicit caches ------------------------------------------------------------------------------
def _dense_fwd(x, w, b):
  return (x @ w + b).astype(F32), (x, w)


def _dense_bwd(dy, cache):
  x, w = cache
  return (dy @ w.T).astype(F32), (x.T @ dy).astype(F32), dy.sum(axis=0).astype(F32)


def _ln_fwd(x, beta, gamma):
  mean = x.mean(axis=1, keepdims=True, dtype=F32)
  var = np.mean(np.square(x - mean), axis=1, keepdims=True, dtype=F32)
  rstd = (F32(1) / np.sqrt(var + LN_EPS)).astype(F32)
  xhat = ((x - mean) * rstd).astype(F32)
  return (xhat * gamma + beta).astype(F32), (xhat, rstd, gamma)


def _ln_bwd(dy, cache):
  xhat, rstd, gamma = cache
  dxhat = dy * gamma
  dx = rstd * (dxhat - dxhat.mean(axis=1, keepdims=True) - xhat * (dxhat * xhat).mean(axis=1, keepdims=True))
  return dx.astype(F32), dy.sum(axis=0).astype(F32), (dy * xhat).sum(axis=0).astype(F32)      # dx, dbeta, dgamma


def _block_fwd(x, p):
  """Dense -> LayerNorm -> ReLU; p = [kernel, bias, beta, gamma] (the reference's creation order)."""
  y, c0 = _dense_fwd(x, p[0], p[1])
  z, c1 = _ln_fwd(y, p[2], p[3])
  return np.maximum(z, F32(0)), (c0, c1, z > 0)


def _block_bwd(dy, cache):
  c0, c1, mask = cache
  dz = dy * mask
  dyl, dbeta, dgamma = _ln_bwd(dz, c1)
  dx, dw, db = _dense_bwd(dyl, c0)
  return dx, [dw, db, dbeta, dgamma]


# -- actor --------------------------------------------------------------------------------------------------------
def actor_forward(params, states, a_min, a_max, depth=2, want_cache=False):
  """params: list of arrays in creation order: depth x [kernel, bias, beta, gamma] + [kernel, bias]."""
  x, caches = np.asarray(states, F32), []
  for d in range(depth):
    x, c = _block_fwd(x, params[4 * d:4 * d + 4])
    caches.append(c)
  u, c = _dense_fwd(x, params[4 * depth], params[4 * depth + 1])
  sig = (F32(1) / (F32(1) + np.exp(-u))).astype(F32)
  a = (sig * F32(a_max - a_min) + F32(a_min)).astype(F32)
  return (a, (caches, c, sig, F32(a_max - a_min))) if want_cache else a


def actor_backward(da, cache):
  caches, c_out, sig, scale = cache
  du = da * scale * sig * (F32(1) - sig)
  dx, dw, db = _dense_bwd(du.astype(F32), c_out)
  grads = [dw, db]
  for c in reversed(caches):
    dx, g = _block_bwd(dx, c)
    grads = g + grads
  return grads


# -- critic -------------------------------------------------------------------------------------------------------
def critic_forward(params, states, actions, depth=2, want_cache=False):
  """params: [kernel, bias, beta, gamma] (state block) + depth x [kernel, bias, beta, gamma] + [kernel, bias]."""
  h, c_s = _block_fwd(np.asarray(states, F32), params[0:4])
  x = np.concatenate([h, np.asarray(actions, F32)], axis=1)
  caches = []
  for d in range(depth):
    x, c = _block_fwd(x, params[4 + 4 * d:8 + 4 * d])
    caches.append(c)
  q, c_out = _dense_fwd(x, params[4 + 4 * depth], params[5 + 4 * depth])
  return (q, (c_s, caches, c_out, h.shape[1])) if want_cache else q


def critic_backward(dq, cache):
  """Returns (parameter gradients in creation order, gradient w.r.t. the action input)."""
  c_s, caches, c_out, width = cache
  dx, dw, db = _dense_bwd(dq.astype(F32), c_out)
  grads = [dw, db]
  for c in reversed(caches):
    dx, g = _block_bwd(dx, c)
    grads = g + grads
  dh, da = dx[:, :width], dx[:, width:]
  _, g = _block_bwd(dh, c_s)
  return g + grads, da.astype(F32)


# -- optimiser / target networks ---------------------------------------------------------------------------------------
class TfAdam(object):
  """tf.train.AdamOptimizer [3P]: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); m, v EMAs; p -= lr_t m / (sqrt(v) + eps)."""

  def __init__(self, params, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
    self.m = [np.zeros_like(p) for p in params]
    self.v = [np.zeros_like(p) for p in params]
    self.t = 0

  def step(self, params, grads):
    self.t += 1
    lr_t = F32(self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t))
    for i, (p, g) in enumerate(zip(params, grads)):
      self.m[i] = (F32(self.b1) * self.m[i] + F32(1.0 - self.b1) * g).astype(F32)
      self.v[i] = (F32(self.b2) * self.v[i] + F32(1.0 - self.b2) * g * g).astype(F32)
      params[i] = (p - lr_t * self.m[i] / (np.sqrt(self.v[i]) + F32(self.eps))).astype(F32)


def soft_update(params_tr, params, tau):
  """agent.py:71-93: var_tr <- (1 - tau) var_tr + tau var (every actor / critic variable is trainable)."""
  return [(F32(1.0 - tau) * pt + F32(tau) * p).astype(F32) for pt, p in zip(params_tr, params)]


class DdpgOracle(object):
  """One `Agent.train()` on a GIVEN mini-batch (sampling and the reward baseline are the caller's business)."""

  def __init__(self, actor, critic, a_min, a_max, depth=2, gamma=0.9, tau=0.01, lr=1e-3, w_dcy=0.0):
    cp = lambda ps: [np.array(p, dtype=F32, copy=True) for p in ps]
    self.actor, self.critic = cp(actor), cp(critic)
    self.actor_tr, self.critic_tr = cp(actor), cp(critic)            # ops['target_init']
    self.a_min, self.a_max, self.depth, self.gamma, self.tau, self.w_dcy = a_min, a_max, depth, gamma, tau, w_dcy
    self.actor_opt, self.critic_opt = TfAdam(self.actor, lr), TfAdam(self.critic, lr)

  def train_on_batch(self, mb):
    s, a, r, term, s2 = (np.asarray(mb[k], F32) for k in ('states', 'actions', 'rewards', 'terminals', 'states_next'))
    n = F32(s.shape[0])
    q_next = critic_forward(self.critic_tr, s2, actor_forward(self.actor_tr, s2, self.a_min, self.a_max, self.depth), self.depth)
    target_q = (r + (F32(1) - term) * F32(self.gamma) * q_next).astype(F32)
    # actor: -mean Q(s, mu(s)); gradient flows through the critic's ACTION input into the actor only
    mu, c_mu = actor_forward(self.actor, s, self.a_min, self.a_max, self.depth, want_cache=True)
    q_mu, c_q = critic_forward(self.critic, s, mu, self.depth, want_cache=True)
    actor_loss = -q_mu.mean(dtype=F32)
    _, dmu = critic_backward(np.full_like(q_mu, -1.0 / n), c_q)
    g_actor = actor_backward(dmu, c_mu)
    # critic: l2_loss(Q(s, a) - target_q) = sum(.)^2 / 2
    q, c = critic_forward(self.critic, s, a, self.depth, want_cache=True)
    diff = (q - target_q).astype(F32)
    critic_loss = F32(np.sum(diff * diff, dtype=F32) / F32(2))
    g_critic, _ = critic_backward(diff, c)
    if self.w_dcy:
      actor_loss += F32(self.w_dcy) * sum(F32(np.sum(p * p) / 2) for p in self.actor)
      critic_loss += F32(self.w_dcy) * sum(F32(np.sum(p * p) / 2) for p in self.critic)
      g_actor = [g + F32(self.w_dcy) * p for g, p in zip(g_actor, self.actor)]
      g_critic = [g + F32(self.w_dcy) * p for g, p in zip(g_critic, self.critic)]
    self.actor_opt.step(self.actor, g_actor)
    self.critic_opt.step(self.critic, g_critic)
    self.actor_tr = soft_update(self.actor_tr, self.actor, self.tau)
    self.critic_tr = soft_update(self.critic_tr, self.critic, self.tau)
    return target_q, float(actor_loss), float(critic_loss)
