"""Hyper-parameter search components against fixtures produced by EXECUTING the reference's own
classes (tests/golden/make_reference_rl_golden.py): actor / critic forward values, replay buffer, noise schedules,
Agent.record / finalize_rlout, the three RL helpers, the channel-pruning reward.  Checked twice: the NumPy oracle
(oracle/ddpg_oracle.py) and the product (pocketflow_amd/rl_agents, learners/*/rl_helper.py).  No GPU."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
A = np.load(os.path.join(HERE, 'golden', 'reference_rl.npz'))
M = json.load(open(os.path.join(HERE, 'golden', 'reference_rl.json')))


@pytest.fixture
def flags():
  import pocketflow_amd.rl_agents.ddpg.agent  # noqa: F401  (defines the ddpg_* flags)
  import pocketflow_amd.learners.uniform_quantization.learner  # noqa: F401
  import pocketflow_amd.learners.nonuniform_quantization.learner  # noqa: F401
  import pocketflow_amd.learners.weight_sparsification.learner  # noqa: F401
  from pocketflow_amd.flags import FLAGS
  saved = {k: getattr(FLAGS, k) for k in ('ddpg_actor_depth', 'ddpg_actor_width', 'ddpg_critic_depth', 'ddpg_critic_width',
                                          'ddpg_record_step', 'ws_prune_ratio', 'ws_reward_type')}
  yield FLAGS
  for k, v in saved.items():
    setattr(FLAGS, k, v)


# -- actor / critic ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize('case', M['actor_critic'], ids=lambda c: c['name'])
def test_oracle_actor_critic_forward(case):
  from oracle import ddpg_oracle as D
  pre = 'ac/%s/' % case['name']
  actor = [A[pre + 'var/' + n] for n in case['actor_vars']]
  critic = [A[pre + 'var/' + n] for n in case['critic_vars']]
  mu = D.actor_forward(actor, A[pre + 'states'], case['a_min'], case['a_max'], case['depth'])
  np.testing.assert_allclose(mu, A[pre + 'mu'], rtol=2e-6, atol=2e-6)
  np.testing.assert_allclose(D.critic_forward(critic, A[pre + 'states'], A[pre + 'actions'], case['depth']), A[pre + 'q'], rtol=2e-6, atol=2e-6)
  np.testing.assert_allclose(D.critic_forward(critic, A[pre + 'states'], A[pre + 'mu'], case['depth']), A[pre + 'q_mu'], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize('case', M['actor_critic'], ids=lambda c: c['name'])
def test_product_actor_critic_forward_and_variable_order(flags, case):
  from pocketflow_amd.rl_agents.ddpg.actor_critic import Actor, Critic
  flags.ddpg_actor_depth = flags.ddpg_critic_depth = case['depth']
  flags.ddpg_actor_width = flags.ddpg_critic_width = case['width']
  pre = 'ac/%s/' % case['name']
  actor = Actor(case['a_dims'], case['a_min'], case['a_max'], scope='agent/actor_mn', rng=np.random.RandomState(0))
  critic = Critic(scope='agent/critic_mn', rng=np.random.RandomState(0))
  s, a = torch.from_numpy(A[pre + 'states']), torch.from_numpy(A[pre + 'actions'])
  with torch.no_grad():
    actor(s), critic(s, a)
  # creation order and names are the reference's (tf.layers / tf.contrib.layers auto-naming)
  assert actor.var_names == case['actor_vars'] and critic.var_names == case['critic_vars']
  actor.load_numpy({n: A[pre + 'var/' + n] for n in case['actor_vars']})
  critic.load_numpy({n: A[pre + 'var/' + n] for n in case['critic_vars']})
  with torch.no_grad():
    mu = actor(s)
    np.testing.assert_allclose(mu.numpy(), A[pre + 'mu'], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(critic(s, a).numpy(), A[pre + 'q'], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(critic(s, mu, reuse=True).numpy(), A[pre + 'q_mu'], rtol=3e-6, atol=3e-6)
  assert [n for n, v in actor.params.items() if any(v is p for p in actor.perturbable_vars)] == \
      [n for n in case['actor_vars'] if 'LayerNorm' not in n]


# -- replay buffer / noise / agent host methods ----------------------------------------------------------------------
def test_replay_buffer_matches_reference_trace():
  from pocketflow_amd.rl_agents.ddpg.replay_buffer import ReplayBuffer
  buf = ReplayBuffer(3, 2, 10)
  for step, row in enumerate(M['replay'][:-1]):
    buf.append(*[A['replay/append%d/%d' % (step, j)] for j in range(5)])
    assert (buf.idx_smpl, buf.nb_smpls, buf.is_ready()) == (row['idx_smpl'], row['nb_smpls'], row['ready'])
    for k in buf.buffers:
      assert np.array_equal(buf.buffers[k], A['replay/after%d/%s' % (step, k)]), (step, k)
  np.random.seed(7)
  mb = buf.sample(6)
  for k in mb:
    assert np.array_equal(mb[k], A['replay/sample_seed7/' + k])
  buf.reset()
  assert (buf.idx_smpl, buf.nb_smpls, buf.is_ready()) == (0, 0, False)


def test_noise_schedules(flags):
  from pocketflow_amd.rl_agents.ddpg.noise import AdaptiveNoiseSpec, TimeDecayNoiseSpec
  n = M['noise']
  td = TimeDecayNoiseSpec(n['tdecy_nb_rlouts'])
  assert td.decy_rat == n['tdecy_rat']
  seq = []
  for _ in n['tdecy']:
    td.adapt()
    seq.append(td.stdev_curr)
  assert seq == n['tdecy']
  td.reset()
  assert td.stdev_curr == n['tdecy_reset']
  ad = AdaptiveNoiseSpec()
  got = []
  for d in n['adapt_dists']:
    ad.adapt(d)
    got.append(ad.stdev_curr)
  assert got == n['adapt']


def test_agent_record_and_reward_baseline(flags):
  from pocketflow_amd.rl_agents.ddpg.agent import Agent
  ag = Agent(0, 3, 1, 10, 4)
  for rewards, want in zip(([0.5] * 4, [0.7, 0.9], [0.1]), M['agent_reward_ema']):
    ag.finalize_rlout(np.array(rewards))
    assert float(ag.reward_ema) == want
  for i in range(3):
    ag.record(*[A['agent_record/in%d/%d' % (i, j)] for j in range(5)])
  for k, v in ag.memory.buffers.items():
    assert np.array_equal(v, A['agent_record/buffers/' + k]), k
  flags.ddpg_record_step = 2
  ag2 = Agent(0, 3, 1, 10, 4)
  ag2.record(*[A['agent_record2/in/%d' % j] for j in range(5)])
  for k, v in ag2.memory.buffers.items():
    assert np.array_equal(v, A['agent_record2/buffers/' + k]), k


# -- RL helpers --------------------------------------------------------------------------------------------------------
class _V(object):
  def __init__(self, shape):
    self.ref_shape = tuple(shape)


@pytest.mark.parametrize('prefix', ['uql', 'nuql'])
def test_bit_allocation_helper(flags, prefix):
  if prefix == 'uql':
    from pocketflow_amd.learners.uniform_quantization.rl_helper import RLHelper
  else:
    from pocketflow_amd.learners.nonuniform_quantization.rl_helper import RLHelper
  setattr(flags, prefix + '_w_bit_min', 2)
  setattr(flags, prefix + '_w_bit_max', 8)
  vars_ = [_V(k) for k in M['kernels']]
  num_weights = [int(np.prod(k)) for k in M['kernels']]
  helpers = {}
  for row in [r for r in M['bit_helpers'] if r['prefix'] == prefix]:
    h = helpers.setdefault(row['eq_bits'], RLHelper(None, row['total_bits'], num_weights, vars_, random_layers=False))
    assert h.s_dims == row['s_dims']
    assert np.array_equal(h.states, A['%s_helper/eq%d/states' % (prefix, row['eq_bits'])])
    h.reset()
    h.layer_idxs = list(row['order'])
    bits, used = [], []
    for idx, a in zip(row['order'], row['raw']):
      assert np.array_equal(h.calc_state(idx)[0], h.states[idx])
      b = h.calc_w(np.array([[a]]), idx)
      assert b.shape == (1, 1)
      bits.append(float(b[0][0]))
      used.append(float(h.w_bits_used))
    assert bits == row['bits'] and used == row['used'], row
    assert h.calc_reward(0.625).tolist() == row['reward']


def test_bit_allocation_helper_shuffles_layers(flags):
  from pocketflow_amd.learners.uniform_quantization.rl_helper import RLHelper
  import random
  h = RLHelper(None, 100, [4, 5, 6, 7], [_V((2, 2)), _V((1, 5)), _V((1, 1, 2, 3)), _V((7, 1))], random_layers=True)
  random.seed(3)
  h.reset()
  want = list(range(4))
  random.seed(3)
  random.shuffle(want)
  assert h.layer_idxs == want and sorted(h.layer_idxs) == [0, 1, 2, 3]


def test_pruning_ratio_helper(flags):
  from pocketflow_amd.learners.weight_sparsification.rl_helper import RLHelper
  vars_ = [_V(k) for k in M['kernels']]
  for row in M['ws_helper']:
    flags.ws_prune_ratio, flags.ws_reward_type = row['ratio'], row['reward_type']
    h = RLHelper(None, vars_, row['skip'])
    tag = 'ws_helper/r%g_skip%d_%s' % (row['ratio'], int(row['skip']), row['reward_type'])
    assert h.s_dims == row['s_dims']
    assert np.array_equal(h.states, A[tag + '/states_static']) and np.array_equal(h.state_normalizer, A[tag + '/normalizer'])
    states, ratios = [], []
    for idx, a in enumerate(row['actions']):
      states.append(h.calc_state(idx)[0])
      ratios.append(float(h.cvt_action_to_prune_ratio(idx, a)))
    assert np.array_equal(np.stack(states), A['%s/case%d/states' % (tag, row['case'])])
    assert ratios == row['prune_ratios'], row
    assert float(h.calc_overall_prune_ratio()) == row['overall'] and float(h.calc_reward(0.8)) == row['reward']


# -- the agent's update step, pinned by the reference's own Agent class executed over oracle/tf_graph_stub.py ------------------
def _agent_case_arrays(case, tag, key):
  return {n: A['agent/%s/%s/%s' % (case['name'], tag, n)] for n in case['vars'][key]}


@pytest.fixture
def ddpg_flags():
  import pocketflow_amd.rl_agents.ddpg.agent  # noqa: F401
  from pocketflow_amd.flags import FLAGS
  names = ('ddpg_actor_depth', 'ddpg_actor_width', 'ddpg_critic_depth', 'ddpg_critic_width', 'ddpg_tau', 'ddpg_gamma',
           'ddpg_lrn_rate', 'ddpg_loss_w_dcy', 'ddpg_batch_size', 'ddpg_enbl_bsln_func', 'ddpg_noise_type', 'ddpg_noise_prtl')
  saved = {k: getattr(FLAGS, k) for k in names}
  yield FLAGS
  for k, v in saved.items():
    setattr(FLAGS, k, v)


def _set_case_flags(FLAGS, case):
  FLAGS.ddpg_actor_depth = FLAGS.ddpg_critic_depth = case['depth']
  FLAGS.ddpg_actor_width = FLAGS.ddpg_critic_width = case['width']
  FLAGS.ddpg_tau, FLAGS.ddpg_gamma, FLAGS.ddpg_lrn_rate = case['tau'], case['gamma'], case['lrn_rate']
  FLAGS.ddpg_loss_w_dcy, FLAGS.ddpg_batch_size, FLAGS.ddpg_enbl_bsln_func = case['w_dcy'], case['batch'], case['bsln']
  FLAGS.ddpg_noise_type, FLAGS.ddpg_noise_prtl = 'param', 'tdecy'


@pytest.mark.parametrize('case', M['agent_update'], ids=lambda c: c['name'])
def test_product_agent_update_matches_the_executed_reference(ddpg_flags, case):
  """`Agent.train()` of the reference (agent.py:216-247 over the graph built in :249-408), executed by
  tests/golden/make_reference_rl_golden.py, against the product agent fed the SAME initial variables and the SAME
  mini-batches: losses, main networks, target networks after each of three updates."""
  from pocketflow_amd.rl_agents.ddpg.agent import Agent
  _set_case_flags(ddpg_flags, case)
  ag = Agent(5, case['s_dims'], case['a_dims'], 10, 64, case['a_min'], case['a_max'])
  ag.init()
  nets = {'actor_mn': ag.actor, 'actor_tr': ag.actor_tr, 'critic_mn': ag.critic, 'critic_tr': ag.critic_tr}
  for key, net in nets.items():
    assert net.var_names == case['vars'][key]
    net.load_numpy(_agent_case_arrays(case, 'init', key))
  worst = 0.0
  for it, ref in enumerate(case['steps']):
    mb = {k: A['agent/%s/batch%d/%s' % (case['name'], it, k)] for k in ('states', 'actions', 'rewards', 'terminals', 'states_next')}
    _, a_loss, c_loss = ag.train_on_batch(mb)
    assert abs(a_loss - ref['actor_loss']) <= 2e-5 * max(1.0, abs(ref['actor_loss'])), (it, a_loss, ref['actor_loss'])
    assert abs(c_loss - ref['critic_loss']) <= 2e-5 * max(1.0, abs(ref['critic_loss'])), (it, c_loss, ref['critic_loss'])
    for key, net in nets.items():
      want = _agent_case_arrays(case, 'after%d' % it, key)
      for n, p in net.params.items():
        worst = max(worst, float(np.max(np.abs(p.detach().numpy() - want[n]))))
  # Adam moves an element by ~lr = 1e-3 per step; agreement is at float32 round-off of the gradients
  assert worst <= 2e-5, worst


@pytest.mark.parametrize('case', M['agent_update'], ids=lambda c: c['name'])
def test_oracle_agent_update_matches_the_executed_reference(case):
  from oracle.ddpg_oracle import DdpgOracle
  init = {k: _agent_case_arrays(case, 'init', k) for k in case['vars']}
  ora = DdpgOracle(list(init['actor_mn'].values()), list(init['critic_mn'].values()), case['a_min'], case['a_max'],
                   depth=case['depth'], gamma=case['gamma'], tau=case['tau'], lr=case['lrn_rate'], w_dcy=case['w_dcy'])
  worst = 0.0
  for it, ref in enumerate(case['steps']):
    mb = {k: A['agent/%s/batch%d/%s' % (case['name'], it, k)] for k in ('states', 'actions', 'rewards', 'terminals', 'states_next')}
    _, a_loss, c_loss = ora.train_on_batch(mb)
    assert abs(a_loss - ref['actor_loss']) <= 2e-5 * max(1.0, abs(ref['actor_loss']))
    assert abs(c_loss - ref['critic_loss']) <= 2e-5 * max(1.0, abs(ref['critic_loss']))
    for key, got in (('actor_mn', ora.actor), ('critic_mn', ora.critic), ('actor_tr', ora.actor_tr), ('critic_tr', ora.critic_tr)):
      for g, w in zip(got, _agent_case_arrays(case, 'after%d' % it, key).values()):
        worst = max(worst, float(np.max(np.abs(g - w))))
  assert worst <= 2e-5, worst


@pytest.mark.parametrize('case', M['agent_update'], ids=lambda c: c['name'])
def test_product_agent_train_draws_the_reference_minibatches(ddpg_flags, case):
  """The whole host path: record -> finalize_rlout (reward baseline) -> train() three times under the same NumPy seed
  draws the mini-batches the reference drew (baseline already subtracted) and ends in the same variables."""
  from pocketflow_amd.rl_agents.ddpg.agent import Agent
  _set_case_flags(ddpg_flags, case)
  ag = Agent(5, case['s_dims'], case['a_dims'], 10, 64, case['a_min'], case['a_max'])
  ag.init()
  nets = {'actor_mn': ag.actor, 'actor_tr': ag.actor_tr, 'critic_mn': ag.critic, 'critic_tr': ag.critic_tr}
  for key, net in nets.items():
    net.load_numpy(_agent_case_arrays(case, 'init', key))
  trans = [A['agent/%s/transitions/%d' % (case['name'], j)] for j in range(5)]
  ag.record(*trans)
  ag.finalize_rlout(trans[2])
  if case['bsln']:
    assert abs(ag.reward_ema - case['reward_ema']) <= 1e-6
  np.random.seed(case['np_seed'])
  ag.memory.rng = np.random                               # the reference samples with the global NumPy generator
  for it, ref in enumerate(case['steps']):
    a_loss, c_loss, std = ag.train()
    assert abs(a_loss - ref['actor_loss']) <= 2e-5 * max(1.0, abs(ref['actor_loss'])), (it, a_loss, ref)
    assert abs(c_loss - ref['critic_loss']) <= 2e-5 * max(1.0, abs(ref['critic_loss'])), (it, c_loss, ref)
    assert std == ref['noise_std']
  for key, net in nets.items():
    want = _agent_case_arrays(case, 'after2', key)
    for n, p in net.params.items():
      assert float(np.max(np.abs(p.detach().numpy() - want[n]))) <= 2e-5, n
