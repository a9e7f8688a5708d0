"""|hip - oracle| / |update| of every variable, in network order, for one step of the masked MobileNet fine-tune (Momentum, float32).
usage: python tools/gpu/cp_step_profile.py [step]"""
import os, sys, tempfile, pathlib, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from parity_common import run_cp_masked_finetune
from pocketflow_amd.flags import FLAGS
import pocketflow_amd.learners.learner_utils  # noqa
import pocketflow_amd.learners.abstract_learner  # noqa
import pocketflow_amd.nets.mobilenet_at_ilsvrc12  # noqa
import pocketflow_amd.learners.channel_pruning.learner  # noqa
import pocketflow_amd.datasets.abstract_dataset  # noqa
want = int(sys.argv[1]) if len(sys.argv) > 1 else 2
FLAGS.reset()
with tempfile.TemporaryDirectory() as d:
  tmp_path = pathlib.Path(d)
  FLAGS.save_path = str(tmp_path / 'models' / 'model.ckpt')
  FLAGS.save_path_eval = str(tmp_path / 'models_eval' / 'model.ckpt')
  FLAGS.synthetic_pool = 2
  FLAGS.compute_dtype = 'float32'
  rep = []
  run_cp_masked_finetune(FLAGS, tmp_path, 'momentum', steps=3, report=rep)


def order(name):
  m = re.search(r'Conv2d_(\d+)(_depthwise|_pointwise)?', name)
  return (int(m.group(1)) if m else 99, 0 if (m and m.group(2) == '_depthwise') else 1, name)


for s, name, e, u in sorted([r for r in rep if r[0] == want], key=lambda r: order(r[1])):
  print('%-70s %.2e' % (name.replace('model/MobilenetV1/', ''), e / max(u, 1e-30)))
