"""How far can ANY bf16-storage implementation of the fine-tune step be from the float32 reference?  (CPU, oracle only.)

The oracle learner (oracle/learner_oracle.py) is run twice from the same conditioned ResNet-v2-50 state on the same batch:
in float32, and with every tensor the product's bf16 mode stores in HBM rounded to bf16 on the way (convolution outputs,
residual sums, the activated / fake-quantised tensors, the quantised kernels, and the same tensors' gradients in the
backward pass).  The per-variable gradient cosine between the two runs is the noise floor a bf16 parity bar has to respect.
    python tests/bf16_noise_probe.py [image_size] [batch] [a_bits]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import learner_oracle as LO


class _R16(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    return x.bfloat16().float()

  @staticmethod
  def backward(ctx, g):
    return g.bfloat16().float()


def conditioned_resnet50_state(values, images, seed=1234, nb_classes=1001, dataset='ilsvrc_12', resnet_size=50,
                               branch_scale=1.0, dense_scale=1.0):
  """A 'pre-trained-like' state from seeded random weights: BN gamma / beta perturbed away from the symmetric (1, 0)
  initialisation, BN moving statistics := the batch statistics of `images` (one calibration pass with momentum 0), so that
  the frozen teacher (inference-mode BN) sees activations of the scale it normalises with."""
  rng = np.random.RandomState(seed)
  vals = {k: np.array(v, dtype=np.float32, copy=True) for k, v in values.items()}
  for k in vals:
    if k.endswith('/gamma'):
      vals[k] = (1.0 + 0.1 * rng.randn(*vals[k].shape)).astype(np.float32)
    elif k.endswith('/beta'):
      vals[k] = (0.1 * rng.randn(*vals[k].shape) - 0.05).astype(np.float32)
  # damp the residual branches (last convolution of every block) and the classifier: a trained network's logits are O(1)
  # and its blocks are perturbations of the identity; seeded He-initialised kernels give neither
  cfgr = LO.resnet_cfg(dataset, resnet_size)
  per_block = 3 if cfgr['bottleneck'] else 2
  idx, last = 1, []
  for i, nblocks in enumerate(cfgr['block_sizes']):
    for b in range(nblocks):
      idx += (1 if b == 0 else 0) + per_block
      last.append(idx - 1)
  for j in last:
    k = 'model/resnet_model/conv2d%s/kernel' % ('' if j == 0 else '_%d' % j)
    vals[k] = vals[k] * np.float32(branch_scale)
  vals['model/resnet_model/dense/kernel'] = vals['model/resnet_model/dense/kernel'] * np.float32(dense_scale)
  s = LO.Scope(vals, 'model', trainable=False)
  orig = LO.Scope.batch_norm

  def calib(self, x, name, momentum, eps, names=('gamma', 'beta', 'moving_mean', 'moving_variance')):
    return orig(self, x, name, 0.0, eps, names)
  LO.Scope.batch_norm = calib
  try:
    s.training = True
    s._begin()
    with torch.no_grad():
      LO.resnet_v2_forward(s, torch.from_numpy(images).permute(0, 3, 1, 2), LO.resnet_cfg(dataset, resnet_size))
  finally:
    LO.Scope.batch_norm = orig
  return {k: t.detach().numpy().copy() for k, t in s.v.items()}


def moved_student(values, rel=0.02, seed=77):
  """The student some way into its fine-tune: every kernel multiplied by (1 + rel * N(0, 1)) element-wise.  With the
  student still EQUAL to the teacher the distillation gradient is a difference of two nearly identical soft-max vectors,
  i.e. pure quantisation / rounding noise, and no two implementations agree on it."""
  if rel <= 0:
    return dict(values)
  rng = np.random.RandomState(seed)
  out = {}
  for k, v in values.items():
    if k.endswith('/kernel'):
      out[k] = (v * (1.0 + rel * rng.randn(*v.shape))).astype(np.float32)
    else:
      out[k] = v
  return out


def cosines(ga, gb):
  out = {}
  for k in ga:
    a, b = ga[k].reshape(-1).astype(np.float64), gb[k].reshape(-1).astype(np.float64)
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    out[k] = (float(a @ b / (na * nb + 1e-300)), float(np.linalg.norm(a - b) / (nb + 1e-300)))
  return out


def main():
  image_size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
  batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
  a_bits = int(sys.argv[3]) if len(sys.argv) > 3 else 8
  branch_scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
  dense_scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
  student_noise = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
  torch.set_num_threads(16)
  rng = np.random.RandomState(0)
  vals = LO._random_values_resnet('ilsvrc_12', 50, 1001, (image_size, image_size, 3), seed=42)
  images = rng.randn(batch, image_size, image_size, 3).astype(np.float32)
  labels = np.eye(1001, dtype=np.float32)[rng.randint(0, 1001, batch)]
  vals = conditioned_resnet50_state(vals, images, branch_scale=branch_scale, dense_scale=dense_scale)
  cfg = dict(model='resnet', dataset='ilsvrc_12', resnet_size=50, nb_classes=1001, loss_w_dcy=1e-4, enbl_dst=True,
             loss_w_dst=4.0, tempr_dst=4.0, momentum=0.9, image_shape=(image_size, image_size, 3), learner='uniform',
             uql_weight_bits=8, uql_activation_bits=a_bits, uql_use_buckets=False)
  tvals = {'distilled_model/' + '/'.join(k.split('/')[1:]): v for k, v in vals.items()}
  vals = moved_student(vals, student_noise)
  ora = LO.OracleLearner(vals, cfg, lambda step: 1e-5, teacher_values=tvals)
  out32, g32 = ora.compute_grads(images, labels)
  # bf16 storage emulation
  o_conv, o_act, o_qw = LO.Scope.conv2d, LO.Scope.activation, LO.Scope._quant_weight
  LO.Scope.conv2d = lambda self, *a, **k: _R16.apply(o_conv(self, *a, **k))
  LO.Scope.activation = lambda self, *a, **k: _R16.apply(o_act(self, *a, **k))
  LO.Scope._quant_weight = lambda self, w, name: _R16.apply(o_qw(self, w, name))
  try:
    ora16 = LO.OracleLearner(vals, cfg, lambda step: 1e-5, teacher_values=tvals)
    out16, g16 = ora16.compute_grads(images.astype(np.float32), labels)
  finally:
    LO.Scope.conv2d, LO.Scope.activation, LO.Scope._quant_weight = o_conv, o_act, o_qw
  print('loss fp32 %.6f  bf16-storage %.6f  (rel %.2e)' % (out32['loss'], out16['loss'], abs(out16['loss'] - out32['loss']) / abs(out32['loss'])))
  cs = cosines(g16, g32)
  kern = sorted((v[0], k) for k, v in cs.items() if k.endswith('kernel'))
  other = sorted((v[0], k) for k, v in cs.items() if not k.endswith('kernel'))
  print('kernel gradients : min cos %.4f (%s), median %.4f, max rel-L2 %.3f' % (kern[0][0], kern[0][1], kern[len(kern) // 2][0], max(cs[k][1] for _, k in kern)))
  print('gamma/beta/bias  : min cos %.4f (%s), median %.4f' % (other[0][0], other[0][1], other[len(other) // 2][0]))
  allg16 = np.concatenate([g16[k].reshape(-1) for k in g16]); allg32 = np.concatenate([g32[k].reshape(-1) for k in g32])
  print('whole gradient   : cos %.5f' % float(allg16 @ allg32 / np.linalg.norm(allg16) / np.linalg.norm(allg32)))
  for c, k in kern[:8]:
    print('   %.4f %s' % (c, k))


if __name__ == '__main__':
  main()
