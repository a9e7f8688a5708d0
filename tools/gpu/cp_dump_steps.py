"""Variables of the product and of the oracle after every step of the masked MobileNet fine-tune (Momentum, float32; every step from a
common state), saved for offline comparison.  usage: [PF_HIP_LIB=...] python tools/gpu/cp_dump_steps.py <out.npz>"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import parity_common as PC
from pocketflow_amd.flags import FLAGS
import pocketflow_amd.learners.learner_utils  # noqa
import pocketflow_amd.learners.abstract_learner  # noqa
import pocketflow_amd.nets.mobilenet_at_ilsvrc12  # noqa
import pocketflow_amd.learners.channel_pruning.learner  # noqa
import pocketflow_amd.datasets.abstract_dataset  # noqa
out, n = {}, [0]
orig = PC._force_state


def force(learner, ora):
  for k, v in learner.graph.store.export_numpy().items():
    out['hip/%d/%s' % (n[0], k)] = v
  for k, v in ora.export().items():
    out['ora/%d/%s' % (n[0], k)] = v
  n[0] += 1
  return orig(learner, ora)


PC._force_state = force
FLAGS.reset()
with tempfile.TemporaryDirectory() as d:
  tmp_path = pathlib.Path(d)
  FLAGS.save_path = str(tmp_path / 'models' / 'model.ckpt')
  FLAGS.save_path_eval = str(tmp_path / 'models_eval' / 'model.ckpt')
  FLAGS.synthetic_pool = 2
  FLAGS.compute_dtype = 'float32'
  PC.run_cp_masked_finetune(FLAGS, tmp_path, 'momentum', steps=3, report=[])
keep = {k: v for k, v in out.items() if any(t in k for t in ('Conv2d_9_', 'Conv2d_10_', 'Conv2d_11_depthwise'))}
np.savez_compressed(sys.argv[1], **keep)
print('saved %d arrays' % len(keep))
