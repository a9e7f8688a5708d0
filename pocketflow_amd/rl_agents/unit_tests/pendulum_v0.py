"""Pendulum swing-up sanity problem for the DDPG agent (reference rl_agents/unit_tests/pendulum_v0.py:28-132).

The reference drives gym's `Pendulum-v0`; gym is not a dependency here, so the environment's published dynamics are
restated [3P]: state (cos th, sin th, th_dot), torque u in [-2, 2], g = 10, m = l = 1, dt = 0.05,
    th_dot <- clip(th_dot + (-3 g / (2 l) sin(th + pi) + 3 u / (m l^2)) dt, -8, 8);  th <- th + th_dot dt
    reward = -(angle_normalize(th)^2 + 0.1 th_dot^2 + 0.001 u^2)   (computed before the update)
`python -m pocketflow_amd.rl_agents.unit_tests.pendulum_v0 --nb_rlouts 100`."""
import logging
import sys

import numpy as np

from pocketflow_amd.flags import FLAGS
from pocketflow_amd.rl_agents.ddpg.agent import Agent as DdpgAgent
from pocketflow_amd.rl_agents.unit_tests import move_to_target  # noqa: F401  (defines nb_rlouts / rlout_len / nb_rlouts_eval)

log = logging.getLogger('pocketflow_amd')


class PendulumEnv(object):
  max_speed, max_torque, dt, g, m, l = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0

  def __init__(self, rng=None):
    self.rng = rng or np.random
    self.th, self.thdot = 0.0, 0.0

  def _obs(self):
    return np.array([np.cos(self.th), np.sin(self.th), self.thdot])

  def reset(self):
    self.th, self.thdot = self.rng.uniform(-np.pi, np.pi), self.rng.uniform(-1.0, 1.0)
    return self._obs()

  def step(self, action):
    u = float(np.clip(np.asarray(action).ravel()[0], -self.max_torque, self.max_torque))
    angle = ((self.th + np.pi) % (2 * np.pi)) - np.pi
    cost = angle ** 2 + 0.1 * self.thdot ** 2 + 0.001 * u ** 2
    self.thdot = float(np.clip(self.thdot + (-3 * self.g / (2 * self.l) * np.sin(self.th + np.pi)
                                             + 3.0 / (self.m * self.l ** 2) * u) * self.dt, -self.max_speed, self.max_speed))
    self.th = self.th + self.thdot * self.dt
    return self._obs(), -cost, False, {}


def build_env_n_agent(sess=None):
  env = PendulumEnv(sess if isinstance(sess, np.random.RandomState) else None)
  buf_size = int(FLAGS.rlout_len * FLAGS.nb_rlouts * 0.25)
  agent = DdpgAgent(sess, 3, 1, FLAGS.nb_rlouts, buf_size, -env.max_torque, env.max_torque)
  return env, agent


def run_rollout(env, agent, train):
  state = env.reset()
  rewards = np.zeros(FLAGS.rlout_len)
  losses = (0.0, 0.0, 0.0)
  for idx_iter in range(FLAGS.rlout_len):
    action = (agent.actions_noisy if train else agent.actions_clean)(state[None, :])
    state_next, reward, __, __ = env.step(action.ravel())
    if train:
      terminal = np.ones((1, 1)) if idx_iter == FLAGS.rlout_len - 1 else np.zeros((1, 1))
      agent.record(state[None, :], action, reward * np.ones((1, 1)), terminal, state_next[None, :])
      losses = agent.train()
    state = state_next
    rewards[idx_iter] = reward
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
