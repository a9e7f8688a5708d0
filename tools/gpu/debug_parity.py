"""Debug: step-0 logits / loss of the HIP learners vs the CPU oracle learner on ResNet-20, several quant configs."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

def run(kind, **kw):
  from pocketflow_amd.flags import FLAGS
  FLAGS.reset()
  import pocketflow_amd.learners.learner_utils, pocketflow_amd.learners.abstract_learner, pocketflow_amd.learners.distillation_helper  # noqa
  from pocketflow_amd.nets.resnet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  from oracle.learner_oracle import OracleLearner
  tmp = tempfile.mkdtemp()
  FLAGS.save_path = os.path.join(tmp, 'models', 'model.ckpt')
  FLAGS.synthetic_pool = 2; FLAGS.compute_dtype = 'float32'
  FLAGS.batch_size = 16; FLAGS.batch_size_eval = 16; FLAGS.resnet_size = 20; FLAGS.nb_classes = 10
  cfg = dict(model='resnet', dataset='cifar_10', resnet_size=20, nb_classes=10, loss_w_dcy=2e-4, enbl_dst=False,
             momentum=0.9, image_shape=(32, 32, 3))
  if kind == 'uq':
    from pocketflow_amd.learners.uniform_quantization.learner import UniformQuantLearner as L
    FLAGS.uql_save_quant_model_path = os.path.join(tmp, 'uql', 'm.ckpt')
    for k, v in kw.items(): setattr(FLAGS, k, v)
    cfg.update(learner='uniform', **kw)
  elif kind == 'nuq':
    from pocketflow_amd.learners.nonuniform_quantization.learner import NonUniformQuantLearner as L
    FLAGS.nuql_save_quant_model_path = os.path.join(tmp, 'nuql', 'm.ckpt')
    for k, v in kw.items(): setattr(FLAGS, k, v)
    cfg.update(learner='non-uniform', **kw)
  mh = ModelHelper(); create_synthetic_checkpoint(mh)
  lrn = L(None, mh)
  if kind == 'nuq': lrn.init_clusters()
  init = lrn.graph.store.export_numpy()
  ora = OracleLearner({k: v for k, v in init.items() if 'clusters' not in k}, cfg, lrn.lrn_rate)
  images, labels = [t.cpu().numpy() for t in lrn.iter_train.batches[0]]
  # product forward only (training mode), no update
  g = lrn.graph
  x, y = lrn.to_device(*lrn.iter_train.batches[0])
  g.begin_step()
  (lrn.uni_quant if kind == 'uq' else lrn.nonuni_quant).quantize_weights()
  with g.as_default():
    logits = lrn.forward_train(x)
    loss, _ = lrn.calc_loss(y, logits, lrn.trainable_vars)
  lg = logits.detach().float().cpu().numpy()
  xt = torch.from_numpy(images); yt = torch.from_numpy(labels)
  ol = ora._forward(ora.student, xt, True)
  oloss, _, _ = ora._loss(yt, ol, None)
  olg = ol.detach().numpy()
  # quantised weights comparison
  st = g.store
  worst = 0
  qsel = set(i for i, b in enumerate(ora.student.quant.w_bits) if b is not None)
  print('%-4s %-60s loss hip %.6f ora %.6f | max|dlogit| %.3e | ce-part hip %.6f' % (
      kind, kw, float(loss), float(oloss), np.abs(lg - olg).max(), float(loss) - 2e-4 * 0))
  l2_hip = 0.5 * (float(torch.dot(st.w_master[:st.w_decay], st.w_master[:st.w_decay])) + float(torch.dot(st.o_master[:st.o_decay], st.o_master[:st.o_decay])))
  l2_ora = sum(0.5 * float((ora.student.v[n] ** 2).sum()) for n in ora._l2_names()) + sum(0.5 * float((c ** 2).sum()) for c in ora.student.quant.codebooks.values())
  print('     L2 sums: hip %.6f ora %.6f  (x wd = %.6f vs %.6f)' % (l2_hip, l2_ora, 2e-4 * l2_hip, 2e-4 * l2_ora))

run('uq', uql_weight_bits=32, uql_activation_bits=32)
run('uq', uql_weight_bits=8, uql_activation_bits=32)
run('uq', uql_weight_bits=32, uql_activation_bits=8)
run('uq', uql_weight_bits=8, uql_activation_bits=8, uql_use_buckets=True, uql_bucket_type='channel')
run('nuq', nuql_weight_bits=3, nuql_activation_bits=32)
run('nuq', nuql_weight_bits=3, nuql_activation_bits=8)
