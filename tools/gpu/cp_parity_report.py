"""Report: distribution of |hip - oracle| / |update| per variable and step of the channel-pruned Momentum fine-tune
(tests/parity_common.run_cp_masked_finetune with report=...), on a clean and on a NaN-poisoned caching allocator."""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch


def poison():
  small = [torch.full((1 << 16,), float('nan'), device='cuda') for _ in range(256)]
  mid = [torch.full((1 << 20,), float('nan'), device='cuda') for _ in range(64)]
  big = [torch.full((1 << 26,), float('nan'), device='cuda') for _ in range(6)]
  torch.cuda.synchronize()
  del small, mid, big


from parity_common import run_cp_masked_finetune
from pocketflow_amd.flags import FLAGS
import pocketflow_amd.learners.learner_utils  # noqa
import pocketflow_amd.learners.abstract_learner  # noqa
import pocketflow_amd.nets.mobilenet_at_ilsvrc12  # noqa
import pocketflow_amd.learners.channel_pruning.learner  # noqa
import pocketflow_amd.datasets.abstract_dataset  # noqa
for tag in ('clean', 'poisoned', 'clean-again'):
  if tag == 'poisoned':
    poison()
  if tag == 'clean-again':
    torch.cuda.empty_cache()
  FLAGS.reset()
  with tempfile.TemporaryDirectory() as d:
    tmp_path = pathlib.Path(d)
    FLAGS.save_path = str(tmp_path / 'models' / 'model.ckpt')
    FLAGS.save_path_eval = str(tmp_path / 'models_eval' / 'model.ckpt')
    FLAGS.synthetic_pool = 2
    FLAGS.compute_dtype = 'float32'
    rep = []
    try:
      run_cp_masked_finetune(FLAGS, tmp_path, 'momentum', steps=3, report=rep)
    except AssertionError as e:
      print(tag, 'assert:', str(e)[:200])
  for stp in sorted(set(r[0] for r in rep)):
    rr = [r for r in rep if r[0] == stp]
    ratios = np.array([r[2] / max(r[3], 1e-30) for r in rr])
    print(tag, 'step', stp, 'global %.3e median %.3e p90 %.3e max %.3e' % (
        np.sqrt(sum(r[2] ** 2 for r in rr)) / np.sqrt(sum(r[3] ** 2 for r in rr)), np.median(ratios), np.percentile(ratios, 90), ratios.max()))
  worst = sorted(rep, key=lambda r: -r[2] / max(r[3], 1e-30))[:6]
  print(tag, 'variables %d; worst err/upd:' % len(rep))
  for stp, name, err, upd in worst:
    print('   step %d %-58s err %.3e upd %.3e ratio %.2e' % (stp, name, err, upd, err / max(upd, 1e-30)))
