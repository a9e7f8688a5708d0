"""Audit of every pf_bn_finalize call of the masked MobileNet fine-tune (tests/parity_common.run_cp_masked_finetune, Momentum, float32):
the kernel's mean / invstd against a float64 two-pass computation over the tensor itself.  Prints the calls whose error exceeds 1e-5
and, per step, the worst call.  usage: [PF_HIP_LIB=...] python tools/gpu/bn_finalize_audit.py"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from parity_common import run_cp_masked_finetune
from pocketflow_amd.flags import FLAGS
from pocketflow_amd import hip
import pocketflow_amd.graph as G
import pocketflow_amd.learners.learner_utils  # noqa
import pocketflow_amd.learners.abstract_learner  # noqa
import pocketflow_amd.nets.mobilenet_at_ilsvrc12  # noqa
import pocketflow_amd.learners.channel_pruning.learner  # noqa
import pocketflow_amd.datasets.abstract_dataset  # noqa

state = {'x': None, 'n': 0, 'log': []}
orig_stats, orig_fin = G._bn_statistics, hip.bn_finalize


def stats(x, rows, C, graph, st):
  state['x'] = x
  return orig_stats(x, rows, C, graph, st)


def fin(partial, nblk, rows, C, piv, gamma, beta, mm, mv, momentum, eps, training, act, ss, mi, slot):
  orig_fin(partial, nblk, rows, C, piv, gamma, beta, mm, mv, momentum, eps, training, act, ss, mi, slot)
  x = state['x']
  if training and x is not None and x.numel() == rows * C:
    xd = (x.permute(0, 2, 3, 1) if x.dim() == 4 else x).reshape(rows, C).double()     # activations are NCHW views of NHWC memory
    mean = xd.mean(0); var = ((xd - mean) ** 2).mean(0)
    inv = 1.0 / torch.sqrt(var + eps)
    e_inv = ((mi[1].double() - inv).abs() / inv).max()
    e_mean = ((mi[0].double() - mean).abs() * inv).max()          # in units of the channel's standard deviation (eps included)
    c = int(((mi[1].double() - inv).abs() / inv).argmax())
    state['log'].append((state['n'], rows, C, nblk, float(e_inv), float(e_mean), c, float(var[c]), float(mean[c]), float(xd[0, c])))
  state['n'] += 1


G._bn_statistics = stats
hip.bn_finalize = fin
FLAGS.reset()
with tempfile.TemporaryDirectory() as d:
  tmp_path = pathlib.Path(d)
  FLAGS.save_path = str(tmp_path / 'models' / 'model.ckpt')
  FLAGS.save_path_eval = str(tmp_path / 'models_eval' / 'model.ckpt')
  FLAGS.synthetic_pool = 2
  FLAGS.compute_dtype = 'float32'
  rep = []
  run_cp_masked_finetune(FLAGS, tmp_path, 'momentum', steps=3, report=rep)
print('%d audited calls' % len(state['log']))
for r in sorted(state['log'], key=lambda r: -max(r[4], r[5]))[:12]:
  print('call %4d rows %6d C %4d n_blocks %4d: invstd rel err %.2e  mean err/std %.2e  (channel %d: var %.3e mean %.3e pivot %.3e)' % r)
for step in range(3):
  e = [x[2] for x in rep if x[0] == step]; u = [x[3] for x in rep if x[0] == step]
  print('step %d global err/upd %.3e' % (step, np.sqrt(np.sum(np.square(e))) / np.sqrt(np.sum(np.square(u)))))
