"""Flat-buffer optimisers: tf.train.AdamOptimizer / MomentumOptimizer over a VarStore.

One fused HIP launch per flat buffer (pf_adam_flat / pf_momentum_flat) applies, in one pass over
HBM: the 1/world_size scaling of the all-reduced gradient, the coupled L2 term of
ModelHelper.calc_loss (`grad += loss_w_dcy * var`, because the reference puts weight decay INTO the
loss -- nets/resnet_at_ilsvrc12.py:132-135), the binary pruning mask (`grad * mask`,
ws learner.py:314-332, cp learner.py:406-419) and the parameter update.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from pocketflow_amd import hip
from pocketflow_amd.graph import VarStore


class FlatOptimizer(object):
  def __init__(self, store: VarStore, kind: str = 'adam', momentum: float = 0.9, beta1: float = 0.9,
               beta2: float = 0.999, epsilon: float = 1e-8, weight_decay: float = 0.0):
    if kind not in ('adam', 'momentum'):
      raise ValueError('unknown optimizer kind: ' + kind)
    self.store, self.kind = store, kind
    self.momentum, self.beta1, self.beta2, self.epsilon = momentum, beta1, beta2, epsilon
    self.weight_decay = weight_decay
    dev = store.device
    self.slots_w = [torch.zeros_like(store.w_master) for _ in range(2 if kind == 'adam' else 1)]
    self.slots_o = [torch.zeros_like(store.o_master) for _ in range(2 if kind == 'adam' else 1)]
    # TF keeps beta1_power / beta2_power as float32 variables initialised to beta and multiplied
    # by beta after every step
    self.beta1_power = np.float32(beta1)
    self.beta2_power = np.float32(beta2)
    self.w_mask: Optional[torch.Tensor] = None      # flat fp32 {0,1} over the W buffer (WS / CP)
    self.o_mask: Optional[torch.Tensor] = None      # flat fp32 {0,1} over the O buffer (var_list subsets)
    self.g_scale = 1.0

  def state_tensors(self) -> List[torch.Tensor]:
    return self.slots_w + self.slots_o

  def reset_slots(self) -> None:
    """tf.variables_initializer(optimizer.variables()) -- `init_opt_op` of the WS learner."""
    for t in self.state_tensors():
      t.zero_()
    self.beta1_power = np.float32(self.beta1)
    self.beta2_power = np.float32(self.beta2)

  def compute_gradients(self) -> None:
    """Gradients already sit in store.w_grad / store.o_grad after backward (single process)."""
    self.g_scale = 1.0

  def apply_gradients(self, lrn_rate: float) -> None:
    st = self.store
    wd = float(self.weight_decay)
    if self.kind == 'adam':
      b1p, b2p = float(self.beta1_power), float(self.beta2_power)
      if st.w_size:
        hip.adam_flat(st.w_master, st.w_grad, self.slots_w[0], self.slots_w[1], self.w_mask, st.w_decay, wd,
                      self.g_scale, lrn_rate, self.beta1, self.beta2, self.epsilon, b1p, b2p)
      if st.o_size:
        hip.adam_flat(st.o_master, st.o_grad, self.slots_o[0], self.slots_o[1], self.o_mask, st.o_decay, wd,
                      self.g_scale, lrn_rate, self.beta1, self.beta2, self.epsilon, b1p, b2p)
      self.beta1_power = np.float32(self.beta1_power * np.float32(self.beta1))
      self.beta2_power = np.float32(self.beta2_power * np.float32(self.beta2))
    else:
      if st.w_size:
        hip.momentum_flat(st.w_master, st.w_grad, self.slots_w[0], self.w_mask, st.w_decay, wd, self.g_scale,
                          lrn_rate, self.momentum)
      if st.o_size:
        hip.momentum_flat(st.o_master, st.o_grad, self.slots_o[0], self.o_mask, st.o_decay, wd, self.g_scale,
                          lrn_rate, self.momentum)
    st.zero_grad()


class DistributedFlatOptimizer(object):
  """mgw.DistributedOptimizer(optimizer): all-reduce(sum) of the flat gradient buffers, then the
  wrapped optimiser with g_scale = 1 / world_size (Horovod's average).  Masks are applied AFTER the
  reduction, as in the reference (ws learner.py:205-207)."""

  def __init__(self, optimizer: FlatOptimizer):
    self.opt = optimizer

  def __getattr__(self, name):
    return getattr(self.opt, name)

  def compute_gradients(self) -> None:
    st = self.opt.store
    if dist.is_initialized() and dist.get_world_size() > 1:
      handles = []
      if st.w_size:
        handles.append(dist.all_reduce(st.w_grad, op=dist.ReduceOp.SUM, async_op=True))
      if st.o_size:
        handles.append(dist.all_reduce(st.o_grad, op=dist.ReduceOp.SUM, async_op=True))
      for h in handles:
        h.wait()
      self.opt.g_scale = 1.0 / dist.get_world_size()
    else:
      self.opt.g_scale = 1.0

  def apply_gradients(self, lrn_rate: float) -> None:
    self.opt.apply_gradients(lrn_rate)
