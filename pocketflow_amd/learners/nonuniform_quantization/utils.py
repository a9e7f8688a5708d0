"""Non-uniform (codebook) fake-quantisation "graph rewrite"
(reference learners/nonuniform_quantization/utils.py:29-494).

Weights: x_hat = (w - beta) / alpha per tensor / per bucket, nearest entry of a trainable codebook
`clusters` (k = 2**bits points, quantile-initialised), y = alpha * c[j*] * sign(x_hat + 1e-6) + beta;
gradients: straight-through to w, scatter-sum of alpha * g into the codebook (override map
{'Mul': 'Add', 'Sign': 'Identity'}, :305-306, 345-346).  Activations use the UNIFORM quantiser (:80).

One `QuantPlan` drives every tensor in two launches (calibration + assign/lookup); the codebooks are
ordinary variables of the model scope (so they are saved, broadcast, L2-regularised and -- depending on
nuql_opt_mode -- optimised like in the reference) whose flat offsets the plan points at.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from pocketflow_amd import hip
from pocketflow_amd.graph import ActivationOp, Graph, MatmulOp, Variable
from pocketflow_amd.plan import QuantPlan


class NonUniformQuantization:
  # pylint: disable=too-many-instance-attributes
  """Class of non-uniform quantization."""

  def __init__(self, graph: Graph, bucket_size=0, use_buckets=False, init_style='quantile',
               bucket_type='split'):
    self.graph = graph
    self.use_buckets = use_buckets
    self.bucket_size = bucket_size
    self.bucket_type = bucket_type
    self.init_style = init_style
    self.matmul_ops: List[MatmulOp] = []
    self.activation_ops: List[ActivationOp] = []
    self.quantized_matmul_ops: List[MatmulOp] = []
    self.quantized_activation_ops: List[ActivationOp] = []
    self.cluster_vars: Dict[int, Variable] = {}
    self.bucket_storage = 0
    self.plan: QuantPlan = None
    self.__safe_check()
    self.support_act_types = ['Relu', 'Relu6', 'Crelu', 'Elu', 'Selu', 'Softplus', 'Softsign', 'Sigmoid', 'Tanh']
    self.support_mul_types = ['Conv2D', 'MatMul', 'DepthwiseConv2dNative']

  def search_matmul_op(self, quantize_all_layers):
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

  def n_bucket_of(self, var: Variable) -> int:
    if not self.use_buckets:
      return 1
    if self.bucket_type == 'channel':
      return 1 if var.kind == 'depthwise' or var.ref_shape[-1] == 1 else var.ref_shape[-1]
    return -(-var.numel // self.bucket_size)

  def declare_clusters(self, w_bit_dict: Dict[str, int], capacity_bits: int = 0):
    """Create the `clusters` variables (must run before the store is finalized).  Shape [k] without
    buckets, [k, bucket_num] with (:297, :324).

    `capacity_bits` > 0 (bit-width search, `nuql_enbl_rl_agent`): the reference declares `clusters` with
    `validate_shape=False` and re-initialises it with a different k = 2**bits every roll-out (:297, nuq
    bit_optimizer.py:234-235); here the variable is allocated once for 2**capacity_bits rows, the first
    2**bits rows are live and the rest stay exactly zero (no gradient, no weight decay, no update)."""
    for op in self.matmul_ops:
      k = 2 ** max(int(w_bit_dict[op.name]), int(capacity_bits))
      nb = self.n_bucket_of(op.var)
      prefix = '/'.join(op.name.split('/')[1:-1])
      scope = 'nonuniform_bucket_quantize' if self.use_buckets else 'nonuniform_quantize'
      shape = (k, nb) if self.use_buckets else (k,)
      self.cluster_vars[id(op.var)] = self.graph.store.add('%s/%s/clusters' % (prefix, scope), shape, 'other',
                                                           trainable=True, l2=True)

  def insert_quant_op_for_weights(self, w_bit_dict: Dict[str, int]):
    store = self.graph.store
    all_vars = store.matmul_vars
    quant = {id(op.var): int(w_bit_dict[op.name]) for op in self.matmul_ops}
    bits = [quant.get(id(v), 0) for v in all_vars]
    cb_offsets = [self.cluster_vars[id(v)].offset if id(v) in self.cluster_vars else 0 for v in all_vars]
    self.plan = QuantPlan(store.weight_descs(all_vars), bits, self.use_buckets, self.bucket_type,
                          self.bucket_size, store.device, nuq=True, cb_offsets=cb_offsets)
    self._all_vars = all_vars
    self._bits = bits
    self._cb_offsets = cb_offsets
    self.idx_flat = torch.zeros(store.w_master.numel(), dtype=torch.uint8, device=store.device)
    self.quantized_matmul_ops = list(self.matmul_ops)
    self.bucket_storage = self.plan.bucket_storage_bits

  def feed_bits(self, w_bits, a_bits):
    """Per-layer bit widths changed (the reference feeds them through placeholders, nuq learner.py:128-129):
    the segment table is rebuilt (host NumPy, a few kB) because codebook sizes follow the bit widths; the
    caller re-initialises the codebooks (`cluster_init`) afterwards, exactly as the reference must."""
    store = self.graph.store
    quant = {id(op.var): int(b) for op, b in zip(self.matmul_ops, w_bits)}
    bits = [quant.get(id(v), 0) for v in self._all_vars]
    if (bits != self._bits or any(op.bits != int(b) for op, b in zip(self.activation_ops, a_bits))) and getattr(self, 'on_change', None):
      self.on_change()                                 # (a step recorded in a hipGraph carries the widths by value: step_graph.py)
    if bits != self._bits:
      for v, b in zip(self._all_vars, bits):
        if b > 0 and (2 ** b) * self.n_bucket_of(v) > self.cluster_vars[id(v)].numel:
          raise ValueError('%d bits do not fit the codebook declared for %s (%d entries)' % (
              b, v.name, self.cluster_vars[id(v)].numel))
      self.plan = QuantPlan(store.weight_descs(self._all_vars), bits, self.use_buckets, self.bucket_type,
                            self.bucket_size, store.device, nuq=True, cb_offsets=self._cb_offsets)
      self._bits = bits
      self.bucket_storage = self.plan.bucket_storage_bits
    for op, b in zip(self.activation_ops, a_bits):
      op.bits = int(b)

  def insert_quant_op_for_activations(self, act_bit_dict: Dict[str, int]):
    for op in self.activation_ops:
      if op.type not in ('Relu', 'Relu6'):
        raise NotImplementedError("The activation_fn needs to include %s manually" % op.type)
      op.bits = int(act_bit_dict[op.name])
      self.quantized_activation_ops.append(op)

  # -- per step ----------------------------------------------------------------------------------
  def quantize_weights(self):
    st = self.graph.store
    self.plan.nonuniform_quantize(st.w_master, st.w_compute, self.idx_flat, st.o_master)

  def codebook_grads(self):
    """dL/dclusters accumulated straight into the flat gradient buffer of the 'other' trainables."""
    st = self.graph.store
    self.plan.codebook_grad(st.w_grad, self.idx_flat, st.o_grad, zero=False)

  # -- ops['cluster_init'] ------------------------------------------------------------------------
  def cluster_init(self):
    """Initial value of every `clusters` variable from the CURRENT weights (nuq learner.py:128-129):
    quantile style c[i] = percentile(x_hat, (i+1)*100/(k+1)) (nearest rank on a descending sort,
    :349-366) or linspace(0, 1, k) (:368-386, un-bucketed only -- SURVEY A.9-3)."""
    st = self.graph.store
    self.plan.calibrate(st.w_master)
    for s, (v, b) in enumerate(zip(self._all_vars, self._bits)):
      if b <= 0:
        continue
      k = 2 ** b
      cvar = self.cluster_vars[id(v)]
      if self.init_style == 'uniform':
        if self.use_buckets:
          raise ValueError('Unrecognized Initialization Mode.')   # broken call in the reference (:225 vs :368)
        self.__write_codebook(cvar, torch.linspace(0., 1., k, device=st.device))
        continue
      if self.init_style != 'quantile':
        raise ValueError('Unrecognized Initialization Mode.')
      xn = torch.empty(v.numel, dtype=torch.float32, device=st.device)
      hip.seg_normalize(st.w_master, xn, self.plan.segs, s, self.plan.slots)
      if not self.use_buckets or self.plan.n_buckets[s] == 1 and self.bucket_type == 'channel':
        X = xn.view(-1, 1)
      elif self.bucket_type == 'channel':
        X = xn.view(v.ref_shape[-1], -1).t()                     # [h*w*cin, cout]
      else:
        ref = xn.view(v.storage_shape)
        if v.kind == 'conv':
          ref = ref.permute(1, 2, 3, 0)
        elif v.kind == 'dense':
          ref = ref.t()
        elif v.kind == 'depthwise':
          ref = ref.permute(1, 2, 0)
        flat = ref.reshape(-1)
        m = self.plan.n_buckets[s]
        pad = m * self.bucket_size - flat.numel()
        if pad:
          flat = torch.cat([flat, flat[-1:].expand(pad)])
        X = flat.view(self.bucket_size, m)
      srt, _ = torch.sort(X, dim=0, descending=True)
      d = X.shape[0]
      rows = []
      for i in range(k):
        q = np.float64((i + 1) * 100) / np.float64(k + 1)
        rows.append(int(np.clip(np.rint(np.float64(d - 1) * (np.float64(1.0) - q / np.float64(100.0))), 0, d - 1)))
      c = srt[torch.tensor(rows, device=st.device)]               # [k, n_bucket]
      self.__write_codebook(cvar, c)

  @staticmethod
  def __write_codebook(cvar, c):
    """The live k rows first, zeros behind them (only a searched codebook is larger than its k rows)."""
    flat = cvar.master.view(-1)
    flat.zero_()
    flat[:c.numel()].copy_(c.reshape(-1))

  def __safe_check(self):
    if self.bucket_size < 0:
      raise ValueError("Bucket size must be a postive integer")
    if self.bucket_type != 'split' and self.bucket_type != 'channel':
      raise ValueError("Unrecognized bucket type, must be 'weight' or 'channel'.")
