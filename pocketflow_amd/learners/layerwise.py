"""Layer-wise access to a network: forward passes in tap mode (graph.Graph.taps) that stop at a given layer.

Shared by the learners that compare an intermediate tensor of two networks layer by layer (the reference builds
`l2_loss(core_op_full.outputs[0] - core_op_prnd.outputs[0])` per Conv2D / MatMul op: weight_sparsification/
pr_optimizer.py:283-316, channel_pruning_gpu/learner.py:339-354).  A tap records (input, output, producer, residual
sum) of every Conv2D / DepthwiseConv2D (and, with `tap_dense`, Dense) layer; `stop_layer` ends the pass right after
that layer's tap, so a regression step on layer i costs two partial forwards + one convolution backward.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from pocketflow_amd import hip
from pocketflow_amd.graph import TapStop, to_device_images


def forward_tapped(graph, forward_fn, images, stop_layer=None, tap_dense=True, grad=False):
  """Run `forward_fn(images)` on `graph` in tap mode; returns {layer: (input, output, producer, residual sum)}.
  `grad=False`: no autograd graph (inference-mode BN is used by forward_eval networks; a forward_train network
  needs `grad=True` to normalise with batch statistics, see graph.BatchNormAct)."""
  graph.taps, graph.tap_dense, graph.tap_stop = OrderedDict(), tap_dense, stop_layer
  try:
    with (torch.enable_grad() if grad else torch.no_grad()), graph.as_default():
      try:
        forward_fn(to_device_images(images, graph))
      except TapStop:
        pass
    return graph.taps
  finally:
    graph.taps, graph.tap_dense, graph.tap_stop = None, False, None


def layers_of_vars(graph, forward_fn, images, variables, grad=False):
  """The layer objects whose kernels are `variables`, in that order."""
  taps = forward_tapped(graph, forward_fn, images, None, grad=grad)
  by_var = {id(layer.kernel): layer for layer in taps}
  return [by_var[id(v)] for v in variables]


class LayerwiseTuner(object):
  """`get_layerwise_tune_op` of the quantisation learners (reference uniform_quantization/utils.py:136-161; the
  non-uniform copy at nonuniform_quantization/utils.py): for quantised matmul op n

      diff_n = reduce_mean(square(op(x, quantised kernel) - op(x, full-precision kernel)))

  with x the op's input in the quantised training graph, minimised w.r.t. the kernel with its own Adam(1e-3).  The
  gradient reaches the kernel twice: through the full-precision op directly and through the quantised op by the
  straight-through estimator.  With the reference's quantisers (range under stop_gradient, Round -> Identity) the two
  contributions cancel exactly, so the op reports the mismatch and leaves the kernel where it is -- which is what
  the reference's "TODO: working not very well" observes; the mechanics are reproduced as they are.  One step = one tapped partial forward of the quantised network + two executions of
  the op + Adam on that kernel alone (pf_adam_flat on the kernel's slice of the flat buffers)."""

  def __init__(self, graph, forward_train, layers, lrn_rate=1e-3):
    self.graph, self.forward_train, self.layers, self.lrn_rate = graph, forward_train, layers, lrn_rate
    self.slots = {}                      # layer index -> [m, v, beta1_power, beta2_power] (one Adam per layer)

  def step(self, n, images, quantize_weights):
    """Run tune op n once; returns diff_n (before the update)."""
    g, layer = self.graph, self.layers[n]
    var = layer.kernel
    st = g.store
    g.begin_step()
    quantize_weights()
    x = forward_tapped(g, self.forward_train, images, layer, tap_dense=True, grad=True)[layer][0].detach()
    st.zero_grad()
    quant_outputs = layer.plain(x)                                   # kernel.tensor = the fake-quantised compute copy
    base = var.master.detach().to(var.tensor.dtype).clone().requires_grad_(True)      # storage layout, full precision
    if var.kind == 'conv':
      w_fp = base.permute(0, 3, 1, 2)
    elif var.kind == 'depthwise':
      w_fp = base.unsqueeze(1)
    else:
      w_fp = base
    quantised, var.tensor = var.tensor, w_fp
    try:
      fp_outputs = layer.plain(x)
    finally:
      var.tensor = quantised
    d = quant_outputs.float() - fp_outputs.float()
    diff = (d * d).mean()
    diff.backward()
    sl = slice(var.offset, var.offset + var.numel)
    grad = (st.w_grad[sl].float() + base.grad.reshape(-1).float()).contiguous()     # STE path + direct path
    if n not in self.slots:
      self.slots[n] = [torch.zeros(var.numel, dtype=torch.float32, device=grad.device),
                       torch.zeros(var.numel, dtype=torch.float32, device=grad.device), np.float32(0.9), np.float32(0.999)]
    m, v, b1p, b2p = self.slots[n]
    hip.adam_flat(st.w_master[sl], grad, m, v, None, 0, 0.0, 1.0, self.lrn_rate, 0.9, 0.999, 1e-8, float(b1p), float(b2p))
    self.slots[n][2], self.slots[n][3] = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
    st.zero_grad()
    return float(diff.detach())
