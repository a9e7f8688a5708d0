"""Host-side launch plan for the segment (all-weights-in-one-launch) kernels.

The reference rewrites the TF graph once per learner (`insert_quant_op_for_weights`,
learners/uniform_quantization/utils.py:81-113) and then feeds the per-layer bit widths through a
placeholder every step (uq learner.py:130-131).  Here the "rewrite" is a table: one 64-byte PfSeg
per weight tensor inside the flat fp32 master buffer plus a block -> (tensor, chunk) map, built
once with NumPy and kept on the device; changing bit widths only rewrites the `bits` column.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from pocketflow_amd import hip


@dataclass
class WeightDesc:
  """One conv / dense / depthwise kernel inside the flat buffers."""
  name: str
  offset: int          # first element in the flat buffers (multiple of 64)
  RS: int              # kh * kw (dense: 1)
  I: int               # reference cin (depthwise: channels)
  O: int               # reference cout (depthwise: channel multiplier = 1)
  layout: int = 0      # 0 = KRSC storage, 1 = CRS (depthwise)

  @property
  def numel(self) -> int:
    return self.RS * self.I * self.O


class QuantPlan:
  """Segment table + block maps + slot/codebook offsets for one set of weight tensors."""

  def __init__(self, weights: Sequence[WeightDesc], bits: Sequence[int], use_buckets: bool,
               bucket_type: str, bucket_size: int, device, nuq: bool = False,
               cb_offsets: Optional[Sequence[int]] = None):
    assert len(weights) == len(bits)
    self.weights = list(weights)
    self.device = device
    self.nuq = nuq
    n = len(weights)
    segs = np.zeros(n, dtype=hip.SEG_DTYPE)
    mm_blocks: List[tuple] = []
    ap_blocks: List[tuple] = []
    slot_off, cb_off = 0, 0
    self.slot_offsets, self.cb_offsets, self.n_buckets = [], [], []
    for s, (w, b) in enumerate(zip(weights, bits)):
      length = w.numel
      if not use_buckets:
        mode, n_bucket = hip.PF_BUCKET_TENSOR, 1
      elif bucket_type == 'channel':
        # __channel_bucket: reshape(w, [-1, cout]); depthwise kernels have cout (multiplier) == 1
        if w.O == 1 or w.layout == 1:
          mode, n_bucket = hip.PF_BUCKET_TENSOR, 1
        else:
          mode, n_bucket = hip.PF_BUCKET_CHANNEL, w.O
      elif bucket_type == 'split':
        mode, n_bucket = hip.PF_BUCKET_SPLIT, -(-length // bucket_size)
      else:
        raise ValueError("Unrecognized bucket type, must be 'split' or 'channel'.")
      if cb_offsets is not None:
        cb_off = int(cb_offsets[s])              # codebooks live in an external buffer (the VarStore)
      segs[s] = (w.offset, length, w.RS, w.layout, w.I, w.O, mode, int(b), bucket_size, n_bucket,
                 slot_off, cb_off)
      self.slot_offsets.append(slot_off)
      self.cb_offsets.append(cb_off)
      self.n_buckets.append(n_bucket)
      n_chunks = -(-length // hip.PF_CHUNK)
      L = w.RS * w.I
      if mode == hip.PF_BUCKET_CHANNEL:
        mm_blocks += [(s, c, 0, 1) for c in range(-(-w.O // 4))]
        for c in range(n_chunks):
          e0 = c * hip.PF_CHUNK
          e1 = min(e0 + hip.PF_CHUNK, length) - 1
          ap_blocks.append((s, c, e0 // L, e1 // L - e0 // L + 1))
      else:
        mm_blocks += [(s, c, 0, 1) for c in range(n_chunks)]
        ap_blocks += [(s, c, 0, 1) for c in range(n_chunks)]
      slot_off += n_bucket
      if nuq and b > 0:
        cb_off += (2 ** int(b)) * n_bucket
    self.n_slots = slot_off
    self.n_codebook = cb_off
    self.segs_host = segs
    self.use_buckets = use_buckets
    self.bucket_type = bucket_type
    self.bucket_size = bucket_size
    self.n_mm_blocks = len(mm_blocks)
    self.n_ap_blocks = len(ap_blocks)
    self.segs = self._to_dev(segs)
    self.mm_blocks = self._to_dev(np.array(mm_blocks, dtype=np.int32).reshape(-1, 4).view(hip.BLOCK_DTYPE))
    self.ap_blocks = self._to_dev(np.array(ap_blocks, dtype=np.int32).reshape(-1, 4).view(hip.BLOCK_DTYPE))
    self.slots = torch.empty((max(self.n_slots, 1), 2), dtype=torch.int32, device=device)
    # bucket storage accounting, uq utils.py:299-306: n_bucket * 32 * 2 bits per bucketed tensor
    self.bucket_storage_bits = sum(nb * 32 * 2 for nb, b in zip(self.n_buckets, bits) if use_buckets and b > 0)

  def _to_dev(self, arr: np.ndarray) -> torch.Tensor:
    raw = np.frombuffer(arr.tobytes(), dtype=np.uint8)
    return torch.from_numpy(raw.copy()).to(self.device)

  def set_bits(self, bits: Sequence[int]) -> None:
    """Per-layer bit widths changed (the reference feeds them through a placeholder each step)."""
    if self.nuq:
      raise ValueError('NUQ codebook sizes depend on the bit widths: rebuild the plan instead')
    self.segs_host['bits'] = np.asarray(bits, dtype=np.int32)
    self.segs.copy_(self._to_dev(self.segs_host), non_blocking=True)

  # -- launches ---------------------------------------------------------------------------------
  def calibrate(self, w_flat: torch.Tensor) -> None:
    """K1 over all tensors: fills self.slots with encoded (min, max) per bucket."""
    hip.minmax_slots_init(self.slots)
    hip.seg_minmax(w_flat, self.segs, self.mm_blocks, self.n_mm_blocks, self.slots)

  def uniform_quantize(self, w_flat: torch.Tensor, qw_flat: torch.Tensor) -> None:
    """K1+K2+K3: qw_flat[seg] = fake_quant(w_flat[seg]) for every tensor (2 launches + 1 memset)."""
    self.calibrate(w_flat)
    hip.seg_uq_apply(w_flat, qw_flat, self.segs, self.ap_blocks, self.n_ap_blocks, self.slots)

  def nonuniform_quantize(self, w_flat, qw_flat, idx_flat, codebooks) -> None:
    """K1+K5: nearest-codebook fake quantisation of every tensor."""
    self.calibrate(w_flat)
    hip.seg_nuq_apply(w_flat, qw_flat, idx_flat, codebooks, self.segs, self.ap_blocks, self.n_ap_blocks,
                      self.slots)

  def codebook_grad(self, g_flat, idx_flat, dcodebooks, zero: bool = True) -> None:
    if zero:
      dcodebooks.zero_()
    ws = getattr(self, '_cb_acc', None)
    if ws is None or ws.numel() < dcodebooks.numel() or ws.device != dcodebooks.device:
      ws = self._cb_acc = torch.empty(dcodebooks.numel(), dtype=torch.int64, device=dcodebooks.device)
    hip.seg_nuq_codebook_grad(g_flat, idx_flat, dcodebooks, ws, self.segs, self.ap_blocks, self.n_ap_blocks,
                              self.slots)

  def alpha_beta(self) -> torch.Tensor:
    """Decoded (alpha, beta) pairs [n_slots, 2] (logging / tests)."""
    return hip.minmax_decode(self.slots)
