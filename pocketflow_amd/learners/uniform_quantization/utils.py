"""Uniform fake-quantisation "graph rewrite" (reference learners/uniform_quantization/utils.py:30-306).

The reference scans the TF graph for Conv2D / MatMul / DepthwiseConv2dNative and activation ops,
builds a chain of ~10 TF ops per tensor and reroutes consumers with graph_editor.  Here the scan walks
the explicit op lists of a `Graph`, and the rewrite is data, not graph surgery:
  * weights      -> one `QuantPlan` (segment table) over the flat fp32 master buffer; every step TWO
                    kernel launches (min/max calibration, fake-quant) produce the compute copy all
                    convolutions read -- K1+K2+K3 of SURVEY section 2.2;
  * activations  -> a bit width on the activation op; the fused BN+ReLU+quant / ReLU+quant kernels
                    do the rest (K4).
Gradients are straight-through (`gradient_override_map({'Round': 'Identity'})`, :185): the gradient of
the loss w.r.t. the quantised compute copy IS the gradient w.r.t. the master weight, so the flat
gradient buffer feeds the optimiser directly.
"""
from __future__ import annotations

from typing import Dict, List

from pocketflow_amd.graph import ActivationOp, Graph, MatmulOp
from pocketflow_amd.plan import QuantPlan


def prefix_filter(prefix):
  """Only keep the '/'-joined scope of an op name (reference :23-28)."""
  return '/'.join(prefix.split('/')[:-1])


class UniformQuantization:
  # pylint: disable=too-many-instance-attributes
  """Class of uniform quantization."""

  def __init__(self, graph: Graph, bucket_size=0, use_buckets=False, bucket_type='split'):
    self.graph = graph                      # (the reference takes a tf.Session; we take the Graph)
    self.use_buckets = use_buckets
    self.bucket_size = bucket_size
    self.bucket_type = bucket_type
    self.matmul_ops: List[MatmulOp] = []
    self.activation_ops: List[ActivationOp] = []
    self.quantized_matmul_ops: List[MatmulOp] = []
    self.quantized_activation_ops: List[ActivationOp] = []
    self.bucket_storage = 0  # bits
    self.plan: QuantPlan = None
    self.__safe_check()
    self.support_act_types = ['Relu', 'Relu6', 'Crelu', 'Elu', 'Selu', 'Softplus', 'Softsign', 'Sigmoid', 'Tanh']
    self.support_mul_types = ['Conv2D', 'MatMul', 'DepthwiseConv2dNative']

  # -- graph scan -------------------------------------------------------------------------------
  def search_matmul_op(self, quantize_all_layers):
    """Matmul ops in creation order, student only; first & last stay full precision (:115-125)."""
    is_student_fn = lambda x: 'distilled' not in x.name
    for op in self.graph.matmul_ops:
      if op.type in self.support_mul_types and is_student_fn(op):
        self.matmul_ops.append(op)
    if not quantize_all_layers:
      self.matmul_ops = self.matmul_ops[1:-1]
    return self.matmul_ops

  def search_activation_op(self):
    is_student_fn = lambda x: 'distilled' not in x.name
    for op in self.graph.activation_ops:
      if op.type in self.support_act_types and is_student_fn(op):
        self.activation_ops.append(op)
    return self.activation_ops

  # -- rewrite ----------------------------------------------------------------------------------
  def insert_quant_op_for_weights(self, w_bit_dict: Dict[str, int]):
    """Build the segment table: quantised tensors get their bit width, all others bits = 0 (cast)."""
    store = self.graph.store
    all_vars = store.matmul_vars
    quant = {id(op.var): w_bit_dict[op.name] for op in self.matmul_ops}
    bits = [int(quant.get(id(v), 0)) for v in all_vars]
    self.plan = QuantPlan(store.weight_descs(all_vars), bits, self.use_buckets, self.bucket_type,
                          self.bucket_size, store.device)
    self._all_vars = all_vars
    self.quantized_matmul_ops = list(self.matmul_ops)
    self.bucket_storage = self.plan.bucket_storage_bits

  def insert_quant_op_for_activations(self, act_bit_dict: Dict[str, int]):
    for op in self.activation_ops:
      if op.type not in ('Relu', 'Relu6'):
        raise NotImplementedError("The activation_fn needs to include %s manually" % op.type)
      op.bits = int(act_bit_dict[op.name])
      self.quantized_activation_ops.append(op)

  def feed_bits(self, w_bits, a_bits):
    """The reference feeds bit widths through int64 placeholders every step (uq learner.py:130-131)."""
    quant = {id(op.var): int(b) for op, b in zip(self.matmul_ops, w_bits)}
    self.plan.set_bits([quant.get(id(v), 0) for v in self._all_vars])
    for op, b in zip(self.activation_ops, a_bits):
      op.bits = int(b)

  def quantize_weights(self):
    """Per step: compute copy <- fake_quant(master) for every matmul kernel (2 launches + 1 memset)."""
    st = self.graph.store
    self.plan.uniform_quantize(st.w_master, st.w_compute)

  def __safe_check(self):
    if self.bucket_size < 0:
      raise ValueError("Bucket size must be a postive integer")
    if self.bucket_type != 'split' and self.bucket_type != 'channel':
      raise ValueError("Unrecognized bucket type, must be 'weight' or 'channel'.")
