"""Loss ops of the hot path as autograd functions over the fused HIP loss kernel.

`softmax_cross_entropy(labels, logits)` == tf.losses.softmax_cross_entropy (mean over the batch of
-sum_c labels*log_softmax(logits)); `distillation_loss` == DistillationHelper.calc_loss
(learners/distillation_helper.py:86-103).  Forward and backward come out of ONE kernel launch
(pf_ce_distill_fwd_bwd); backward only rescales the stored dlogits by the upstream scalar.
"""
from __future__ import annotations

import torch

from pocketflow_amd import hip


def _run_kernel(z_s, labels, z_t, tempr, loss_w):
  B, C = z_s.shape
  z_s = z_s.contiguous()
  losses = torch.empty(2, dtype=torch.float32, device=z_s.device)
  dz = torch.empty((B, C), dtype=z_s.dtype, device=z_s.device)
  row_ws = torch.empty(B * 2, dtype=torch.float32, device=z_s.device)
  hip.ce_distill_fwd_bwd(z_s, labels, z_t, tempr, loss_w, losses, dz, row_ws)
  return losses, dz


class _SoftmaxCE(torch.autograd.Function):
  @staticmethod
  def forward(ctx, logits, labels):
    losses, dz = _run_kernel(logits, labels.contiguous().float(), None, 1.0, 0.0)
    ctx.save_for_backward(dz)
    return losses[0]

  @staticmethod
  def backward(ctx, g):
    (dz,) = ctx.saved_tensors
    return dz * g.to(dz.dtype), None


class _DistillCE(torch.autograd.Function):
  @staticmethod
  def forward(ctx, logits_pri, logits_dst, tempr, loss_w):
    zeros = torch.zeros(logits_pri.shape, dtype=torch.float32, device=logits_pri.device)
    losses, dz = _run_kernel(logits_pri, zeros, logits_dst.contiguous(), float(tempr), float(loss_w))
    ctx.save_for_backward(dz)
    return losses[1]

  @staticmethod
  def backward(ctx, g):
    (dz,) = ctx.saved_tensors
    return dz * g.to(dz.dtype), None, None, None


def softmax_cross_entropy(labels: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
  return _SoftmaxCE.apply(logits, labels)


def distillation_loss(logits_pri: torch.Tensor, logits_dst: torch.Tensor, tempr: float,
                      loss_w: float) -> torch.Tensor:
  return _DistillCE.apply(logits_pri, logits_dst.detach(), tempr, loss_w)


def l2_regularization(trainable_vars, loss_filter, loss_w_dcy: float) -> torch.Tensor:
  """loss_w_dcy * add_n([tf.nn.l2_loss(v) for v in trainable_vars if loss_filter(v)]).

  The VALUE is returned (for the logged loss); its GRADIENT (wd * v) is applied inside the fused
  optimiser kernel, so the returned tensor is detached.  The filter must agree with the `l2` flag
  each variable was declared with (that flag decides the layout of the flat buffers)."""
  tot = None
  store = None
  for v in trainable_vars:
    want = bool(loss_filter(v))
    if want != bool(v.l2):
      raise ValueError('calc_loss regularises %s but the variable was declared with l2=%s' % (v.name, v.l2))
    store = store or getattr(v, 'store', None)
  store = store or (trainable_vars[0].store if trainable_vars else None)
  if store is None:
    return torch.zeros(())
  store.weight_decay = float(loss_w_dcy)
  w = store.w_master[:store.w_decay]
  o = store.o_master[:store.o_decay]
  tot = 0.5 * (torch.dot(w, w) + torch.dot(o, o))
  return (loss_w_dcy * tot).detach()


def in_top_k(outputs: torch.Tensor, targets: torch.Tensor, k: int) -> torch.Tensor:
  """tf.nn.in_top_k: fewer than k entries are strictly greater than the target's score."""
  o = outputs.float()
  t = o.gather(1, targets.view(-1, 1))
  return (o > t).sum(dim=1) < k
