"""Debug: per-activation comparison HIP vs oracle, ResNet-20 UQ w32/a8, step-0 forward."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pocketflow_amd.flags import FLAGS
import pocketflow_amd.learners.learner_utils, pocketflow_amd.learners.abstract_learner, pocketflow_amd.learners.distillation_helper  # noqa
from pocketflow_amd.nets.resnet_at_cifar10 import ModelHelper
from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
from pocketflow_amd.learners.uniform_quantization.learner import UniformQuantLearner as L
from pocketflow_amd import graph as G, hip
from oracle import learner_oracle as LO
tmp = tempfile.mkdtemp()
FLAGS.save_path = os.path.join(tmp, 'models', 'model.ckpt')
FLAGS.synthetic_pool = 2; FLAGS.compute_dtype = 'float32'
FLAGS.batch_size = 16; FLAGS.resnet_size = 20; FLAGS.nb_classes = 10
FLAGS.uql_save_quant_model_path = os.path.join(tmp, 'uql', 'm.ckpt')
FLAGS.uql_weight_bits = 32; FLAGS.uql_activation_bits = 8
cfg = dict(model='resnet', dataset='cifar_10', resnet_size=20, nb_classes=10, loss_w_dcy=2e-4, enbl_dst=False,
           momentum=0.9, image_shape=(32, 32, 3), learner='uniform', uql_weight_bits=32, uql_activation_bits=8)
mh = ModelHelper(); create_synthetic_checkpoint(mh)
lrn = L(None, mh)
init = lrn.graph.store.export_numpy()
ora = LO.OracleLearner(init, cfg, lrn.lrn_rate)
hip_outs, hip_ins = [], []
orig = G.BatchNormAct.__call__
def rec(self, x):
  q = orig(self, x)
  hip_ins.append(x.detach().float().cpu().numpy()); hip_outs.append(q.detach().float().cpu().numpy())
  return q
G.BatchNormAct.__call__ = rec
ora_outs, ora_ins = [], []
oact = LO.Scope.activation
def orec(self, u, kind, name):
  o = oact(self, u, kind, name)
  ora_ins.append(u.detach().numpy()); ora_outs.append(o.detach().numpy())
  return o
LO.Scope.activation = orec
g = lrn.graph
x, y = lrn.to_device(*lrn.iter_train.batches[0])
g.begin_step()
lrn.uni_quant.quantize_weights()
with g.as_default():
  logits = lrn.forward_train(x)
ab = hip.minmax_decode(g.act_slots).cpu().numpy()
images = lrn.iter_train.batches[0][0].cpu().numpy()
ol = ora._forward(ora.student, torch.from_numpy(images), True)
for i, (hq, oq, hx, ou) in enumerate(zip(hip_outs, ora_outs, hip_ins, ora_ins)):
  t = np.maximum(ou, 0)
  print('act %2d shape %-18s |dq|max %.3e  n_diff %6d/%d | hip x vs ora(pre-bn n/a) | hip alpha,beta %.6f %.6f  ora max,min %.6f %.6f  uniq hip %d ora %d' % (
      i, hq.shape, np.abs(hq - oq).max(), int((np.abs(hq - oq) > 1e-6).sum()), hq.size, ab[i, 0], ab[i, 1], t.max(), t.min(),
      len(np.unique(hq)), len(np.unique(oq))))
  if i >= 5: break
