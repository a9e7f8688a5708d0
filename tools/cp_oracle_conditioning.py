"""CPU: conditioning of the masked MobileNet fine-tune instance of tests/parity_common.run_cp_masked_finetune (Momentum).  Per step:
(a) |emulated product - oracle| / |update| (HIP entry points emulated in float32 with torch on the CPU: a THIRD summation order),
(b) |oracle on images moved by one float32 rounding - oracle| / |update|: the oracle against itself."""
import os, sys, tempfile, pathlib
from unittest import mock
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from fake_hip import FakeHipFull
import pocketflow_amd.graph as G
import pocketflow_amd.plan as P
import pocketflow_amd.losses as L
import pocketflow_amd.optim as Opt
import pocketflow_amd.learners.abstract_learner as AL
import pocketflow_amd.learners.weight_sparsification.learner as WS
import pocketflow_amd.learners.nonuniform_quantization.utils as NU
import pocketflow_amd.learners.layerwise as LW
import pocketflow_amd.learners.distillation_helper  # noqa
import pocketflow_amd.learners.learner_utils  # noqa
import pocketflow_amd.nets.mobilenet_at_ilsvrc12  # noqa
import pocketflow_amd.learners.channel_pruning.learner as CP
from pocketflow_amd.flags import FLAGS
from parity_common import run_cp_masked_finetune

fake = FakeHipFull()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with tempfile.TemporaryDirectory() as d, mock.patch.object(AL, 'require_gpu', lambda: torch.device('cpu')), \
     mock.patch.object(G, 'DEPTHWISE_ANY_DEVICE', True):
  for mod in (G, P, L, Opt, WS, NU, LW, CP):
    mod.hip = fake
  tmp = pathlib.Path(d)
  FLAGS.save_path = str(tmp / 'models' / 'model.ckpt')
  FLAGS.save_path_eval = str(tmp / 'models_eval' / 'model.ckpt')
  FLAGS.synthetic_pool, FLAGS.compute_dtype = 2, 'float32'
  rep, cond = [], []
  run_cp_masked_finetune(FLAGS, tmp, 'momentum', steps=steps, report=rep, conditioning=cond)
for what, rows in (('emulated product vs oracle', rep), ('oracle vs its twins (images moved by TWIN_JITTER * N(0, 1))', cond)):
  for step in range(steps):
    r = np.array([e / max(u, 1e-30) for s, n, e, u in rows if s == step and u > 0])
    E = np.sqrt(sum(e * e for s, n, e, u in rows if s == step)); U = np.sqrt(sum(u * u for s, n, e, u in rows if s == step))
    print('%-50s step %d: global %.3e  median %.3e  p90 %.3e  max %.3e' % (what, step, E / U, np.median(r), np.percentile(r, 90), r.max()))
