#!/usr/bin/env python
"""Reference-executed fixture for the 'chn-pruned-gpu' proximal step.

Runs HERE (needs /root/reference); the GPU box only reads the committed output (reference_cpg.npz / .json).

The step is not a function of its own in the reference: it is the five statements under the `tf.control_dependencies` block of
ChannelPrunedGpuLearner.__build_layer_ops (learners/channel_pruning_gpu/learner.py:379-383).  They are located in the parsed
file by their assignment targets (no source is copied), compiled as they stand and EXECUTED over oracle/tf_stub.py with the
free names they use bound to values: `var_prnd` a variable stand-in with `.assign`, `grads` = [(gradient, var_prnd)],
`lrn_rate_pgd` / `prune_perctl` float32 scalars (placeholders in the reference, fed per layer at learner.py:452-456).

What this pins: the order of the operations, the axes and keepdims of the norm, that the percentile is taken over the
per-input-channel norms with the contrib default ('nearest'), the shrink expression.  What it does not pin: the summation order
inside tf.reduce_sum (the stub sums with NumPy; TensorFlow's Eigen reduction may differ in the last bit) -- the cases are sized
so that the test is meaningful either way, and tests/test_oracle_golden.py states it."""
import ast
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_reference_golden as G  # noqa: E402  (the tf stub with its FLAGS, REF)
from oracle import tf_stub  # noqa: E402

PATH = 'learners/channel_pruning_gpu/learner.py'
TARGETS = ['var_prnd_new', 'var_norm', 'threshold', 'shrk_vec', 'prune_op']

# (kernel shape HWIO, weight scale, gradient scale, learning rate, percentile)
CASES = [
    ((3, 3, 8, 16), 0.1, 0.05, 0.01, 30.0),
    ((3, 3, 8, 16), 0.1, 0.05, 0.01, 0.0),          # threshold = the smallest norm: exactly one channel is zeroed
    ((3, 3, 8, 16), 0.1, 0.05, 0.01, 100.0),        # threshold = the largest norm: every channel is zeroed
    ((1, 1, 32, 64), 0.05, 0.5, 0.1, 50.0),
    ((1, 1, 32, 64), 0.05, 0.5, 0.1, 12.5),         # (d - 1) * (1 - q / 100) = 27.125: the 'nearest' rounding
    ((5, 5, 3, 24), 0.2, 1.0, 1e-3, 66.0),
    ((3, 3, 32, 48), 0.03, 0.02, 0.05, 37.0),
    ((7, 7, 3, 64), 0.1, 0.3, 0.02, 34.0),          # (3 - 1) * 0.66 = 1.32
    ((1, 1, 6, 4), 1.0, 1.0, 0.25, 50.0),           # (6 - 1) * 0.5 = 2.5: round half to even
]


class Var(tf_stub.T):
  """The slice of tf.Variable the statements touch."""

  def assign(self, value, **kw):
    self.a = np.asarray(tf_stub._raw(value))
    return self


def lifted_block():
  """The statements of __build_layer_ops that assign TARGETS, in file order, compiled as one module."""
  tree = ast.parse(open(os.path.join(G.REF, PATH)).read())
  cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'ChannelPrunedGpuLearner'][0]
  fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name.endswith('__build_layer_ops')][0]
  found = []
  for node in ast.walk(fn):
    if isinstance(node, ast.With):
      names = [s.targets[0].id for s in node.body
               if isinstance(s, ast.Assign) and len(s.targets) == 1 and isinstance(s.targets[0], ast.Name)]
      if names == TARGETS:
        found.append(node)
  assert len(found) == 1, 'the proximal block moved: %d matches' % len(found)
  lines = (found[0].body[0].lineno, found[0].body[-1].end_lineno)
  mod = ast.Module(body=found[0].body, type_ignores=[])
  ast.fix_missing_locations(mod)
  return compile(mod, os.path.join(G.REF, PATH), 'exec'), lines


def main():
  code, lines = lifted_block()
  out, meta = {}, {'reference': '%s:%d-%d' % (PATH, lines[0], lines[1]), 'cases': []}
  rng = np.random.RandomState(20240917)
  for i, (shape, ws, gs, lr, q) in enumerate(CASES):
    w = (rng.randn(*shape) * ws).astype(np.float32)
    g = (rng.randn(*shape) * gs).astype(np.float32)
    var = Var(w.copy())
    ns = {'tf': G.tf, 'var_prnd': var, 'grads': [(tf_stub.T(g), var)],
          'lrn_rate_pgd': tf_stub.T(np.float32(lr)), 'prune_perctl': tf_stub.T(np.float32(q))}
    exec(code, ns)
    assert ns['prune_op'] is var
    new = np.asarray(var.a)
    norm = np.asarray(tf_stub._raw(ns['var_norm']))
    thr = np.asarray(tf_stub._raw(ns['threshold']))
    assert new.dtype == np.float32 and norm.dtype == np.float32 and norm.shape == (1, 1, shape[2], 1), (new.dtype, norm.shape)
    k = 'cpg/%d/' % i
    out[k + 'w'], out[k + 'g'], out[k + 'new'], out[k + 'norm'], out[k + 'thr'] = w, g, new, norm.reshape(-1), np.float32(thr)
    zeroed = int(np.sum(np.all(new == 0, axis=(0, 1, 3))))
    meta['cases'].append({'shape': list(shape), 'lrn_rate': lr, 'prune_perctl': q, 'channels_zeroed': zeroed})
  np.savez_compressed(os.path.join(HERE, 'reference_cpg.npz'), **out)
  with open(os.path.join(HERE, 'reference_cpg.json'), 'w') as f:
    json.dump(meta, f, indent=1, sort_keys=True)
    f.write('\n')
  print('reference_cpg: %d cases from %s; channels zeroed %s' % (len(CASES), meta['reference'], [c['channels_zeroed'] for c in meta['cases']]))


if __name__ == '__main__':
  main()
