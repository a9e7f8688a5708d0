"""Replay buffer of the DDPG agent (reference rl_agents/ddpg/replay_buffer.py:21-121).

A host-side float32 ring of (state, action, reward, terminal, next state) rows: the roll-outs that fill it are
a few dozen rows long (one row per layer of the compressed network), so it never leaves the host.
Same contract as the reference: `is_ready()` only once the ring is FULL, `sample()` draws with replacement."""
import numpy as np

_KEYS = ('states', 'actions', 'rewards', 'terminals', 'states_next')


class ReplayBuffer(object):
  def __init__(self, s_dims, a_dims, buf_size, rng=None):
    self.s_dims, self.a_dims, self.buf_size = s_dims, a_dims, int(buf_size)
    self.rng = rng                                  # None: the global NumPy generator, as in the reference
    self.idx_smpl = 0
    self.nb_smpls = 0
    widths = {'states': s_dims, 'actions': a_dims, 'rewards': 1, 'terminals': 1, 'states_next': s_dims}
    self.buffers = {k: np.zeros((self.buf_size, widths[k]), dtype=np.float32) for k in _KEYS}

  def reset(self):
    self.idx_smpl = 0
    self.nb_smpls = 0

  def is_ready(self):
    return self.nb_smpls == self.buf_size

  def append(self, states, actions, rewards, terminals, states_next):
    rows = dict(zip(_KEYS, (states, actions, rewards, terminals, states_next)))
    n = np.asarray(states).shape[0]
    # ring positions of the n new rows (the reference splits into a tail and a head copy, :95-105)
    pos = (self.idx_smpl + np.arange(n)) % self.buf_size
    for k in _KEYS:
      self.buffers[k][pos] = np.asarray(rows[k], dtype=np.float32).reshape(n, -1)
    self.idx_smpl = (self.idx_smpl + n) % self.buf_size
    self.nb_smpls = min(self.nb_smpls + n, self.buf_size)

  def sample(self, batch_size):
    draw = (self.rng or np.random).randint
    idxs = draw(0, self.nb_smpls, batch_size)
    return {k: self.buffers[k][idxs] for k in _KEYS}
