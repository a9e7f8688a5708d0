#!/usr/bin/env python
"""Generate tests/golden/reference_rl.npz / reference_rl.json by EXECUTING THE REFERENCE'S OWN hyper-parameter
search components.  Same method and caveats as make_reference_golden.py: definitions are lifted
with `ast` from the files under /root/reference and run over oracle/tf_stub.py; build container only.

    python tests/golden/make_reference_rl_golden.py

Lifted (paths relative to /root/reference):
  rl_agents/ddpg/actor_critic.py                 dense_block, Model, Actor, Critic  (forward values for given variables)
  rl_agents/ddpg/replay_buffer.py                ReplayBuffer   (pure NumPy)
  rl_agents/ddpg/noise.py                        AdaptiveNoiseSpec, TimeDecayNoiseSpec
  rl_agents/ddpg/agent.py                        Agent.finalize_rlout, Agent.record (the two methods without graph code)
  learners/uniform_quantization/rl_helper.py     RLHelper
  learners/nonuniform_quantization/rl_helper.py  RLHelper
  learners/weight_sparsification/rl_helper.py    RLHelper
  learners/channel_pruning/learner.py            ChannelPrunedLearner.__calc_reward
  learners/channel_pruning/channel_pruner.py     ChannelPruner.initialize_state, getState, __action_constraint, __conv_left,
                                                 __compute_model_flops, finallayer  (RL state table, strategy table and the
                                                 FLOP-target action constraint, run on a stand-in model wrapper that
                                                 describes two small topologies; pandas is the real one)
  rl_agents/ddpg/agent.py                        Agent (whole class: __build, init, record, finalize_rlout, train) over the
                                                 deferred-execution stand-in oracle/tf_graph_stub.py -> the update step
Not executable here (TF graph construction / sessions inside learner loops): BitOptimizer, PROptimizer roll-out loops;
they are restated in pocketflow_amd and anchored on the pieces above.
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_reference_golden as G  # noqa: E402  (installs the stub, provides lift())

tf, FLAGS, T = G.tf, G.FLAGS, G.tf_stub.T

DDPG_DEFAULTS = dict(ddpg_actor_depth=2, ddpg_actor_width=64, ddpg_critic_depth=2, ddpg_critic_width=64,
                     ddpg_noise_type='param', ddpg_noise_prtl='tdecy', ddpg_noise_std_init=1e+0, ddpg_noise_dst_finl=1e-2,
                     ddpg_noise_adpt_rat=1.03, ddpg_noise_std_finl=1e-5, ddpg_record_step=1, ddpg_enbl_bsln_func=True,
                     ddpg_bsln_decy_rate=0.95)


def set_flags(**kw):
  for k, v in kw.items():
    setattr(FLAGS, k, v)


def net_variables(rng, scope, widths, ln_after):
  """Random variables of a dense(+LayerNorm) stack: widths = [(in, out), ...]; ln_after[i] -> LayerNorm behind dense i."""
  vals, k_dense, k_ln = {}, 0, 0
  for (cin, cout), ln in zip(widths, ln_after):
    d = 'dense' if k_dense == 0 else 'dense_%d' % k_dense
    vals['%s/%s/kernel' % (scope, d)] = (rng.randn(cin, cout) / np.sqrt(cin)).astype(np.float32)
    vals['%s/%s/bias' % (scope, d)] = (0.1 * rng.randn(cout)).astype(np.float32)
    k_dense += 1
    if ln:
      l = 'LayerNorm' if k_ln == 0 else 'LayerNorm_%d' % k_ln
      vals['%s/%s/beta' % (scope, l)] = (0.1 * rng.randn(cout)).astype(np.float32)
      vals['%s/%s/gamma' % (scope, l)] = (1.0 + 0.1 * rng.randn(cout)).astype(np.float32)
      k_ln += 1
  return vals


def gen_actor_critic(out, meta):
  ns = G.lift('rl_agents/ddpg/actor_critic.py', ['dense_block', 'Model', 'Actor', 'Critic'], {'ENBL_LAYER_NORM': True})
  rng = np.random.RandomState(31)
  cases = []
  for name, s_dims, a_dims, a_min, a_max, width, depth, batch in (
      ('bits', 9, 1, 0.0, 6.0, 64, 2, 1), ('move', 4, 4, -1.0, 1.0, 64, 2, 7), ('narrow', 6, 2, 0.2, 1.0, 16, 3, 5)):
    set_flags(ddpg_actor_depth=depth, ddpg_actor_width=width, ddpg_critic_depth=depth, ddpg_critic_width=width)
    a_scope, c_scope = 'agent/actor_mn', 'agent/critic_mn'
    a_vals = net_variables(rng, a_scope, [(s_dims, width)] + [(width, width)] * (depth - 1) + [(width, a_dims)],
                           [True] * depth + [False])
    c_vals = net_variables(rng, c_scope, [(s_dims, width), (width + a_dims, width)] + [(width, width)] * (depth - 1) + [(width, 1)],
                           [True] * (depth + 1) + [False])
    states = rng.randn(batch, s_dims).astype(np.float32)
    actions = rng.uniform(a_min, a_max, (batch, a_dims)).astype(np.float32)
    G.tf_stub.reset_layers(dict(a_vals, **c_vals))
    actor, critic = ns['Actor'](a_dims, a_min, a_max, scope=a_scope), ns['Critic'](scope=c_scope)
    mu = actor(T(states))
    a_order = list(G.tf_stub.variables_used)
    q = critic(T(states), T(actions))
    c_order = list(G.tf_stub.variables_used)[len(a_order):]
    q_mu = critic(T(states), mu, reuse=True)                       # second call re-enters the scope: same variables
    assert list(G.tf_stub.variables_used)[len(a_order) + len(c_order):] == c_order
    for k, v in dict(a_vals, **c_vals).items():
      out['ac/%s/var/%s' % (name, k)] = v
    out['ac/%s/states' % name], out['ac/%s/actions' % name] = states, actions
    out['ac/%s/mu' % name], out['ac/%s/q' % name], out['ac/%s/q_mu' % name] = mu.numpy(), q.numpy(), q_mu.numpy()
    cases.append(dict(name=name, s_dims=s_dims, a_dims=a_dims, a_min=a_min, a_max=a_max, width=width, depth=depth,
                      actor_vars=a_order, critic_vars=c_order))
  meta['actor_critic'] = cases
  set_flags(**DDPG_DEFAULTS)


def gen_replay(out, meta):
  RB = G.lift('rl_agents/ddpg/replay_buffer.py', ['ReplayBuffer'])['ReplayBuffer']
  rng = np.random.RandomState(5)
  buf = RB(3, 2, 10)
  trace = []
  for step, n in enumerate((4, 4, 1, 3, 7, 1)):
    rows = [rng.randn(n, 3), rng.randn(n, 2), rng.randn(n, 1), (rng.rand(n, 1) > 0.7).astype(np.float64), rng.randn(n, 3)]
    for j, r in enumerate(rows):
      out['replay/append%d/%d' % (step, j)] = r
    buf.append(*rows)
    trace.append(dict(n=n, idx_smpl=int(buf.idx_smpl), nb_smpls=int(buf.nb_smpls), ready=bool(buf.is_ready())))
    for k, v in buf.buffers.items():
      out['replay/after%d/%s' % (step, k)] = v.copy()
  np.random.seed(7)
  mb = buf.sample(6)
  for k, v in mb.items():
    out['replay/sample_seed7/%s' % k] = v
  buf.reset()
  trace.append(dict(n=0, idx_smpl=int(buf.idx_smpl), nb_smpls=int(buf.nb_smpls), ready=bool(buf.is_ready())))
  meta['replay'] = trace


def gen_noise(meta):
  ns = G.lift('rl_agents/ddpg/noise.py', ['AdaptiveNoiseSpec', 'TimeDecayNoiseSpec'])
  set_flags(**DDPG_DEFAULTS)
  td = ns['TimeDecayNoiseSpec'](20)
  seq = []
  for _ in range(25):
    td.adapt()
    seq.append(td.stdev_curr)
  td.reset()
  ad = ns['AdaptiveNoiseSpec']()
  dists = [0.5, 0.2, 0.005, 0.02, 0.001, 0.0099, 0.0101]
  aseq = []
  for d in dists:
    ad.adapt(d)
    aseq.append(ad.stdev_curr)
  meta['noise'] = dict(tdecy_nb_rlouts=20, tdecy=seq, tdecy_rat=td.decy_rat, tdecy_reset=td.stdev_curr, adapt_dists=dists, adapt=aseq)


def gen_agent_host(out, meta):
  """Agent.finalize_rlout / record do not touch the graph: run them on a shell object."""
  ns = G.lift('rl_agents/ddpg/agent.py', ['Agent.finalize_rlout', 'Agent.record'])
  RB = G.lift('rl_agents/ddpg/replay_buffer.py', ['ReplayBuffer'])['ReplayBuffer']
  set_flags(**DDPG_DEFAULTS)
  ag = ns['Agent']()
  ag.reward_ema, ag.state_rms = None, None
  ema = []
  for rewards in ([0.5] * 4, [0.7, 0.9], [0.1]):
    ag.finalize_rlout(np.array(rewards))
    ema.append(float(ag.reward_ema))
  meta['agent_reward_ema'] = ema
  # record with (1,1)-shaped rewards / terminals, as every learner calls it
  ag.memory = RB(3, 1, 4)
  rng = np.random.RandomState(9)
  for i in range(3):
    s, a, r, t, s2 = rng.randn(1, 3), rng.rand(1, 1), 0.25 * (i + 1) * np.ones((1, 1)), np.float64(i == 2) * np.ones((1, 1)), rng.randn(1, 3)
    for j, x in enumerate((s, a, r, t, s2)):
      out['agent_record/in%d/%d' % (i, j)] = x
    ag.record(s, a, r, t, s2)
  for k, v in ag.memory.buffers.items():
    out['agent_record/buffers/%s' % k] = v.copy()
  # record with 1-D rewards over a multi-row roll-out and ddpg_record_step = 2 (rl_agents/unit_tests usage)
  set_flags(ddpg_record_step=2)
  ag.memory = RB(3, 1, 4)
  s, a, r, t, s2 = rng.randn(5, 3), rng.rand(5, 1), rng.randn(5), np.zeros(5), rng.randn(5, 3)
  for j, x in enumerate((s, a, r, t, s2)):
    out['agent_record2/in/%d' % j] = x
  ag.record(s, a, r, t, s2)
  for k, v in ag.memory.buffers.items():
    out['agent_record2/buffers/%s' % k] = v.copy()
  set_flags(ddpg_record_step=1)


KERNELS = [(3, 3, 3, 8), (3, 3, 8, 32), (1, 1, 32, 64), (3, 3, 64, 1), (64, 10)]


def gen_bit_helpers(out, meta):
  rows = []
  for prefix, path in (('uql', 'learners/uniform_quantization/rl_helper.py'), ('nuql', 'learners/nonuniform_quantization/rl_helper.py')):
    RL = G.lift(path, ['RLHelper'], {'random': random})['RLHelper']
    set_flags(**{prefix + '_w_bit_min': 2, prefix + '_w_bit_max': 8})
    vars_ = [T(np.zeros(s, np.float32)) for s in KERNELS]
    num_weights = [int(np.prod(s)) for s in KERNELS]
    for eq_bits in (4, 3, 7):
      total_bits = sum(num_weights) * eq_bits
      h = RL(tf.Session(), total_bits, num_weights, vars_, random_layers=False)
      out['%s_helper/eq%d/states' % (prefix, eq_bits)] = h.states.copy()
      for case, (order, raw) in enumerate((([0, 1, 2, 3, 4], [5.2, 0.4, 2.5, 3.5, 1.0]), ([3, 1, 4, 0, 2], [6.0, 6.0, 6.0, 6.0, 6.0]),
                                           ([4, 3, 2, 1, 0], [0.0, 1.49, 1.5, 5.9, 0.3]))):
        h.reset()
        h.layer_idxs = list(order)
        bits, used = [], []
        for idx, a in zip(order, raw):
          st = h.calc_state(idx)
          b = h.calc_w(np.array([[a]]), idx)
          bits.append(float(b[0][0]))
          used.append(float(h.w_bits_used))
          assert st.shape == (1, h.s_dims) and b.shape == (1, 1)
        rows.append(dict(prefix=prefix, eq_bits=eq_bits, case=case, order=list(order), raw=list(raw), bits=bits, used=used,
                         total_bits=total_bits, s_dims=int(h.s_dims), reward=h.calc_reward(0.625).tolist()))
  meta['bit_helpers'] = rows
  meta['kernels'] = [list(k) for k in KERNELS]


def gen_ws_helper(out, meta):
  RL = G.lift('learners/weight_sparsification/rl_helper.py', ['RLHelper'])['RLHelper']
  rows = []
  vars_ = [T(np.zeros(s, np.float32)) for s in KERNELS]
  for ratio in (0.75, 0.5, 0.9):
    for skip in (True, False):
      if skip and ratio > 0.8:
        continue                                   # head & tail dense: the target is unreachable on this small net
      for reward_type in ('single-obj', 'multi-obj'):
        set_flags(ws_prune_ratio=ratio, ws_reward_type=reward_type)
        h = RL(tf.Session(), vars_, skip)
        tag = 'ws_helper/r%g_skip%d_%s' % (ratio, int(skip), reward_type)
        out[tag + '/states_static'] = h.states.copy()
        out[tag + '/normalizer'] = h.state_normalizer.copy()
        for case, actions in enumerate(([0.5] * 5, [0.0, 1.0, 0.25, 0.75, 0.6], [1.0, 1.0, 1.0, 1.0, 1.0], [0.1, 0.1, 0.0, 0.3, 0.2])):
          h.prune_ratios[:] = 0
          states, ratios = [], []
          for idx, a in enumerate(actions):
            states.append(h.calc_state(idx)[0])
            ratios.append(float(h.cvt_action_to_prune_ratio(idx, a)))
          out['%s/case%d/states' % (tag, case)] = np.stack(states)
          rows.append(dict(ratio=ratio, skip=skip, reward_type=reward_type, case=case, actions=list(actions), prune_ratios=ratios,
                           overall=float(h.calc_overall_prune_ratio()), reward=float(h.calc_reward(0.8)), s_dims=int(h.s_dims)))
  meta['ws_helper'] = rows


def gen_cp_reward(meta):
  ns = G.lift('learners/channel_pruning/learner.py', ['ChannelPrunedLearner.__calc_reward'])
  fn = getattr(ns['ChannelPrunedLearner'], '_ChannelPrunedLearner__calc_reward')
  rows = []
  for policy in ('accuracy', 'flops'):
    for acc, flops in ((0.9, 0.5), (0.99, 0.25), (0.3, 0.7)):
      set_flags(cp_reward_policy=policy, cp_noise_tolerance=0.15)
      rows.append(dict(policy=policy, acc=acc, flops=flops, reward=np.asarray(fn(ns['ChannelPrunedLearner'], acc, flops)).tolist()))
  meta['cp_reward'] = rows


class _Op(object):
  def __init__(self, name, type_, n=0, c=0, k=1, stride=1, hw=0):
    self.name, self.type, self.n, self.c, self.k, self.stride, self.hw = name, type_, n, c, k, stride, hw
    self.flops = 2.0 * hw * hw * k * k * c * n


class _Wrapper(object):
  """Stand-in for learners/channel_pruning/model_wrapper.py:Model with a hand-written topology."""

  def __init__(self, ops, fathers):
    self.ops, self.fathers, self.g = ops, fathers, self
  def get_operation_by_name(self, name): return [o for o in self.ops if o.name == name][0]
  def get_operations_by_type(self, otype='Conv2D'): return [o for o in self.ops if o.type == otype]
  def is_W1_prunable(self, conv): return self.fathers[conv.name] is not None
  def get_conv_def(self, op): return {'n': op.n, 'c': op.c, 'h': op.k, 'w': op.k, 'strides': [1, op.stride, op.stride, 1]}
  def get_outname_by_opname(self, name): return name
  def output_width(self, name): return self.get_operation_by_name(name).hw
  def output_height(self, name): return self.get_operation_by_name(name).hw
  def compute_layer_flops(self, op): return op.flops


CP_TOPOLOGIES = {
    # MobileNet-like chain: conv, (depthwise, pointwise) x 2, 1x1 classifier behind a global pool
    'chain': ([('conv0', 'Conv2D', 8, 3, 3, 2, 16), ('dw1', 'DepthwiseConv2dNative', 8, 8, 3, 1, 16), ('pw1', 'Conv2D', 16, 8, 1, 1, 16),
               ('dw2', 'DepthwiseConv2dNative', 16, 16, 3, 2, 8), ('pw2', 'Conv2D', 32, 16, 1, 1, 8), ('fc', 'Conv2D', 10, 32, 1, 1, 1)],
              {'conv0': None, 'dw1': 'conv0', 'pw1': 'dw1', 'dw2': 'pw1', 'pw2': 'dw2', 'fc': None}),
    # pre-activation residual net: stem, block 1 (identity shortcut), block 2 (1x1 stride-2 projection shortcut);
    # the convolutions fed by a residual sum (b2p, b2c1) have no single producer
    'resnet': ([('stem', 'Conv2D', 8, 3, 3, 1, 16), ('b1c1', 'Conv2D', 8, 8, 3, 1, 16), ('b1c2', 'Conv2D', 8, 8, 3, 1, 16),
                ('b2p', 'Conv2D', 16, 8, 1, 2, 8), ('b2c1', 'Conv2D', 16, 8, 3, 2, 8), ('b2c2', 'Conv2D', 16, 16, 3, 1, 8)],
               {'stem': None, 'b1c1': 'stem', 'b1c2': 'b1c1', 'b2p': None, 'b2c1': None, 'b2c2': 'b2c1'}),
}


def gen_cp_states(out, meta):
  import math
  import pandas as pd
  ns = G.lift('learners/channel_pruning/channel_pruner.py',
              ['ChannelPruner.initialize_state', 'ChannelPruner.getState', 'ChannelPruner.__action_constraint',
               'ChannelPruner.__conv_left', 'ChannelPruner.__compute_model_flops', 'ChannelPruner.finallayer'],
              {'pd': pd, 'math': math})
  CP = ns['ChannelPruner']
  rows = []
  for topo, (ops, fathers) in CP_TOPOLOGIES.items():
    for preserve, policy in ((0.5, 'accuracy'), (0.3, 'accuracy'), (0.5, 'flops')):
      set_flags(cp_preserve_ratio=preserve, cp_reward_policy=policy, cp_prune_option='auto')
      model = _Wrapper([_Op(*o) for o in ops], fathers)
      pr = CP()
      pr._model, pr.lbound, pr.state, pr.drop_conv = model, math.log(preserve + 1, 10) * 1.5, 0, set([])
      pr.thisconvs = model.get_operations_by_type()
      pr.initialize_state()
      tag = 'cp_states/%s_p%g_%s' % (topo, preserve, policy)
      out[tag + '/states'] = pr.states.values.astype(np.float64)
      strategy0 = {k: list(v) for k, v in pr.max_strategy_dict.items()}
      for case, actions in enumerate(([0.9, 0.8, 0.7, 0.6, 0.5, 0.4], [0.25, 0.3, 0.35, 1.0, 0.2, 0.9], [1.5, -0.2, 0.45, 0.45, 0.45, 0.45])):
        pr.initialize_state()
        got, maxred = [], []
        for i, conv in enumerate(pr.thisconvs):
          a = actions[i]
          if pr.state == 0:
            a = 1.0
          if pr.finallayer():
            a = 1
          c = pr._ChannelPruner__action_constraint(a)
          got.append(float(c))
          maxred.append(float(pr.max_reduced_flops))
          # the bookkeeping compress() / prune_W1 / prune_W2 do with the ratio that was applied (:665-770)
          father = model.fathers[conv.name]
          if c == 1:
            pr.max_strategy_dict[conv.name][0] = c
            if father is not None and father in pr.max_strategy_dict:
              pr.max_strategy_dict[father][1] = c
          else:
            pr.max_strategy_dict[conv.name][0] = c
            while father is not None and model.get_operation_by_name(father).type == 'DepthwiseConv2dNative' and model.fathers[father] is not None:
              father = model.fathers[father]
            if father is not None and father in pr.max_strategy_dict:
              pr.max_strategy_dict[father][1] = c
          if not pr.finallayer():
            pr.state += 1
            pr.currentStates['maxreduce'][pr.state] = pr.max_reduced_flops / pr.model_flops
        out['%s/case%d/current_states' % (tag, case)] = pr.currentStates.values.astype(np.float64)
        rows.append(dict(topo=topo, preserve=preserve, policy=policy, case=case, actions=list(actions), constrained=got,
                         max_reduced_flops=maxred, model_flops=float(pr.model_flops), lbound=float(pr.lbound),
                         desired_preserve=float(pr.desired_preserve), strategy0=strategy0,
                         pruned_flops=float(pr._ChannelPruner__compute_model_flops(fake=True))))
  meta['cp_states'] = rows
  meta['cp_topologies'] = {k: {'ops': [list(o) for o in v[0]], 'fathers': v[1]} for k, v in CP_TOPOLOGIES.items()}


def gen_agent_update(out, meta):
  """The reference's `Agent.__build / init / record / finalize_rlout / train` EXECUTED as they are written, over the
  deferred-execution stand-in oracle/tf_graph_stub.py (placeholders, optimizer.minimize, tf.assign, sess.run): initial
  variables, the mini-batches `train()` drew, and every variable (main + target networks, Adam slots) after each of three
  updates.  This is what pins the update step of the product agent and of oracle/ddpg_oracle.py."""
  sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
  import oracle.tf_graph_stub as tfg
  cases = []
  for name, s_dims, a_dims, a_min, a_max, width, depth, w_dcy, bsln, batch in (
      ('default', 5, 2, 0.0, 1.0, 64, 2, 0.0, True, 16), ('decay_narrow', 7, 3, -1.0, 2.0, 16, 3, 1e-3, False, 8)):
    set_flags(**DDPG_DEFAULTS)
    set_flags(ddpg_actor_depth=depth, ddpg_actor_width=width, ddpg_critic_depth=depth, ddpg_critic_width=width,
              ddpg_tau=0.01, ddpg_gamma=0.9, ddpg_lrn_rate=1e-3, ddpg_loss_w_dcy=w_dcy, ddpg_batch_size=batch,
              ddpg_enbl_bsln_func=bsln)
    tfg.reset_default_graph()
    tfg.seed(11 + len(cases))
    ac = G.lift('rl_agents/ddpg/actor_critic.py', ['dense_block', 'Model', 'Actor', 'Critic'], {'tf': tfg, 'ENBL_LAYER_NORM': True})
    rb = G.lift('rl_agents/ddpg/replay_buffer.py', ['ReplayBuffer'])
    nz = G.lift('rl_agents/ddpg/noise.py', ['AdaptiveNoiseSpec', 'TimeDecayNoiseSpec'])
    ag_ns = G.lift('rl_agents/ddpg/agent.py',
                   ['normalize', 'denormalize', 'calc_loss_dcy', 'get_target_model_ops', 'get_perturb_op', 'Agent'],
                   {'tf': tfg, 'Actor': ac['Actor'], 'Critic': ac['Critic'], 'ReplayBuffer': rb['ReplayBuffer'],
                    'AdaptiveNoiseSpec': nz['AdaptiveNoiseSpec'], 'TimeDecayNoiseSpec': nz['TimeDecayNoiseSpec'],
                    'RunningMeanStd': None})
    agent = ag_ns['Agent'](tfg.Session(), s_dims, a_dims, 10, 64, a_min, a_max)
    agent.init()
    nets = {'actor_mn': agent.actor, 'actor_tr': agent.actor_tr, 'critic_mn': agent.critic, 'critic_tr': agent.critic_tr}
    model_vars = {k: [v for v in tfg.get_collection(tfg.GraphKeys.TRAINABLE_VARIABLES, scope=m.scope)] for k, m in nets.items()}

    def snapshot(tag):
      for k, vs in model_vars.items():
        for v in vs:
          out['agent/%s/%s/%s' % (name, tag, v.name[:-2])] = v.value.numpy().copy()
    snapshot('init')
    # the target networks start as copies of the main networks (ops['target_init'])
    for a, b in ((agent.actor, agent.actor_tr), (agent.critic, agent.critic_tr)):
      for va, vb in zip(model_vars['actor_mn' if a is agent.actor else 'critic_mn'], model_vars['actor_tr' if a is agent.actor else 'critic_tr']):
        assert np.array_equal(va.value.numpy(), vb.value.numpy())
    rng = np.random.RandomState(40 + len(cases))
    n = 70                                              # > buf_size: the buffer is "ready" only when full (and wraps once)
    trans = [rng.randn(n, s_dims).astype(np.float32), rng.uniform(a_min, a_max, (n, a_dims)).astype(np.float32),
             rng.randn(n).astype(np.float32), (rng.rand(n) > 0.8).astype(np.float32), rng.randn(n, s_dims).astype(np.float32)]
    for j, t in enumerate(trans):
      out['agent/%s/transitions/%d' % (name, j)] = t
    agent.record(*trans)
    agent.finalize_rlout(trans[2])
    drawn = []
    sample = agent.memory.sample

    def logging_sample(batch_size):
      mb = sample(batch_size)
      drawn.append(mb)
      return mb
    agent.memory.sample = logging_sample
    np.random.seed(3 + len(cases))
    steps = []
    for it in range(3):
      a_loss, c_loss, std = agent.train()
      steps.append(dict(actor_loss=float(a_loss), critic_loss=float(c_loss), noise_std=float(std)))
      snapshot('after%d' % it)
      for k, v in drawn[-1].items():                      # as FED: the baseline is already subtracted from the rewards
        out['agent/%s/batch%d/%s' % (name, it, k)] = np.asarray(v, dtype=np.float32).copy()
    cases.append(dict(name=name, s_dims=s_dims, a_dims=a_dims, a_min=a_min, a_max=a_max, width=width, depth=depth,
                      w_dcy=w_dcy, bsln=bsln, batch=batch, tau=0.01, gamma=0.9, lrn_rate=1e-3, np_seed=3 + len(cases),
                      reward_ema=None if agent.reward_ema is None else float(agent.reward_ema), steps=steps,
                      vars={k: [v.name[:-2] for v in vs] for k, vs in model_vars.items()}))
  meta['agent_update'] = cases
  set_flags(**DDPG_DEFAULTS)


def main():
  arrays, meta = {}, {}
  gen_actor_critic(arrays, meta)
  gen_agent_update(arrays, meta)
  gen_replay(arrays, meta)
  gen_noise(meta)
  gen_agent_host(arrays, meta)
  gen_bit_helpers(arrays, meta)
  gen_ws_helper(arrays, meta)
  gen_cp_reward(meta)
  gen_cp_states(arrays, meta)
  np.savez_compressed(os.path.join(HERE, 'reference_rl.npz'), **arrays)
  with open(os.path.join(HERE, 'reference_rl.json'), 'w') as f:
    json.dump(meta, f, indent=1, sort_keys=True)
  print('wrote %d arrays (%.1f KiB) and %d host tables' % (
      len(arrays), os.path.getsize(os.path.join(HERE, 'reference_rl.npz')) / 1024.0, len(meta)))


if __name__ == '__main__':
  main()
