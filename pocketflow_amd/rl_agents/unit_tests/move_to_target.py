"""Move-to-target sanity problem for the DDPG agent (reference rl_agents/unit_tests/move_to_target.py:32-168).

A point in [-10, 10]^d moves by the action each tick; reward = Dist(x, 0) - Dist(x', 0) - Dist(x, x') <= 0 with
equality for straight moves towards the origin.  `python -m pocketflow_amd.rl_agents.unit_tests.move_to_target`."""
import logging
import sys

import numpy as np
from numpy.linalg import norm

from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.rl_agents.ddpg.agent import Agent as DdpgAgent

flags.DEFINE_integer('nb_dims', 4, '# of state & action dimensions')
flags.DEFINE_integer('nb_rlouts', 200, '# of roll-outs')
flags.DEFINE_integer('nb_rlouts_eval', 100, '# of roll-outs for evaluation')
flags.DEFINE_integer('rlout_len', 200, 'roll-out\'s length')

log = logging.getLogger('pocketflow_amd')


class Env(object):
  def __init__(self, rng=None):
    self.rng = rng or np.random
    self.x_lbnd, self.x_ubnd = -10.0, 10.0
    self.x_curr = None
    self.target = np.zeros((1, FLAGS.nb_dims))

  def reset(self):
    self.x_curr = self.rng.uniform(self.x_lbnd, self.x_ubnd, (1, FLAGS.nb_dims))
    return self.x_curr

  def step(self, action):
    x_next = self.x_curr + action
    reward = norm(self.x_curr - self.target) - norm(x_next - self.target) - norm(self.x_curr - x_next)
    self.x_curr = x_next
    return self.x_curr, reward * np.ones((1, 1))


def build_env_n_agent(sess=None):
  env = Env(sess if isinstance(sess, np.random.RandomState) else None)
  buf_size = int(FLAGS.rlout_len * FLAGS.nb_rlouts * 0.25)
  return env, DdpgAgent(sess, FLAGS.nb_dims, FLAGS.nb_dims, FLAGS.nb_rlouts, buf_size, -1.0, 1.0)


def run_rollout(env, agent, train):
  state = env.reset()
  rewards = np.zeros(FLAGS.rlout_len)
  losses = (0.0, 0.0, 0.0)
  for idx_iter in range(FLAGS.rlout_len):
    action = agent.actions_noisy(state) if train else agent.actions_clean(state)
    state_next, reward = env.step(action)
    if train:
      terminal = np.ones((1, 1)) if (idx_iter == FLAGS.rlout_len - 1) else np.zeros((1, 1))
      agent.record(state, action, reward, terminal, state_next)
      losses = agent.train()
    state = state_next
    rewards[idx_iter] = reward[0, 0]
  return rewards, losses


def train_agent(env, agent):
  agent.init()
  history = []
  for idx_rlout in range(FLAGS.nb_rlouts):
    agent.init_rlout()
    rewards, (actor_loss, critic_loss, noise_std) = run_rollout(env, agent, train=True)
    agent.finalize_rlout(rewards)
    history.append(float(np.mean(rewards)))
    log.info('roll-out #%d: reward (ave.): %.2e | a-loss = %.2e | c-loss = %.2e | noise std. = %.2e',
             idx_rlout, history[-1], actor_loss, critic_loss, noise_std)
  return history


def eval_agent(env, agent):
  means = [float(np.mean(run_rollout(env, agent, train=False)[0])) for _ in range(FLAGS.nb_rlouts_eval)]
  log.info('[EVAL] reward (ave.): %.4e', np.mean(means))
  return float(np.mean(means))


def main(argv=None):
  FLAGS.parse(argv if argv is not None else sys.argv[1:])
  logging.basicConfig(level=logging.INFO)
  env, agent = build_env_n_agent(None)
  train_agent(env, agent)
  eval_agent(env, agent)
  return 0


if __name__ == '__main__':
  sys.exit(main())
