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
    self.w_grad_src: Optional[torch.Tensor] = None  # set by the distributed wrapper: reduced gradients (staging)
    self.o_grad_src: Optional[torch.Tensor] = None
    # captured steps (step_graph.py): per-step scalars in device memory, hp = [alpha_t (Adam), lr, 0, 0]
    self.hp: Optional[torch.Tensor] = None
    self.hyper_external = False

  def enable_device_hyper(self) -> None:
    """From now on the update kernels read the learning rate (Adam: alpha_t) from device memory, written by `feed_hyper` before
    every step: what a step recorded in a hipGraph needs (kernel arguments passed by value are frozen at capture).  The values are
    the float32 numbers the by-value path computes -- bit-identical updates."""
    if self.hp is None:
      self.hp = torch.zeros(4, dtype=torch.float32, device=self.store.device)

  def feed_hyper(self, lrn_rate: float) -> None:
    """Write this step's scalars (one tiny launch; values travel as kernel arguments) and advance the beta powers."""
    one = np.float32(1.0)
    alpha_t = 0.0
    if self.kind == 'adam':
      # TF ApplyAdam, float32, in the evaluation order of pf_adam_flat: (lr * sqrt(1 - b2^t)) / (1 - b1^t)
      alpha_t = float(np.float32(np.float32(lrn_rate) * np.sqrt(one - np.float32(self.beta2_power))) / (one - np.float32(self.beta1_power)))
      self.beta1_power = np.float32(self.beta1_power * np.float32(self.beta1))
      self.beta2_power = np.float32(self.beta2_power * np.float32(self.beta2))
    hip.set_floats(self.hp, alpha_t, float(lrn_rate))

  def state_tensors(self) -> List[torch.Tensor]:
    return self.slots_w + self.slots_o

  def reset_slots(self) -> None:
    """tf.variables_initializer(optimizer.variables()) -- `init_opt_op` of the WS learner."""
    for t in self.state_tensors():
      t.zero_()
    self.beta1_power = np.float32(self.beta1)
    self.beta2_power = np.float32(self.beta2)

  def backward(self, loss: torch.Tensor) -> None:
    """The backward pass of a training step.  Learners call this (not `loss.backward()`) for the backward that is
    followed by compute_gradients() / apply_gradients(): the distributed wrapper overrides it to let the gradient
    exchange start from inside the pass.  Any OTHER backward (layer-wise tuning on rank 0, regression-gradient
    helpers) stays a plain `loss.backward()` and never touches the process group.  The backward-filter launches of the pass run on a
    second queue (graph.WrwSide)."""
    from pocketflow_amd import graph as G
    with G.wrw_side_armed(self.store):
      loss.backward()

  def compute_gradients(self) -> None:
    """Gradients already sit in store.w_grad / store.o_grad after backward (single process)."""
    self.g_scale = 1.0
    self.w_grad_src = self.o_grad_src = None

  def apply_gradients(self, lrn_rate: float) -> None:
    st = self.store
    wd = float(self.weight_decay)
    w_grad = self.w_grad_src if self.w_grad_src is not None else st.w_grad
    o_grad = self.o_grad_src if self.o_grad_src is not None else st.o_grad
    if self.hp is not None:
      # device-scalar mode.  `hyper_external`: a StepGraph drives this optimiser -- it feeds before every replay, and while the
      # step is being recorded nothing must be fed (a capture executes nothing)
      if not self.hyper_external:
        self.feed_hyper(lrn_rate)
      if self.kind == 'adam':
        if st.w_size:
          hip.adam_flat_dev(st.w_master, w_grad, self.slots_w[0], self.slots_w[1], self.w_mask, st.w_decay, wd, self.g_scale,
                            self.hp, self.beta1, self.beta2, self.epsilon)
        if st.o_size:
          hip.adam_flat_dev(st.o_master, o_grad, self.slots_o[0], self.slots_o[1], self.o_mask, st.o_decay, wd, self.g_scale,
                            self.hp, self.beta1, self.beta2, self.epsilon)
      else:
        if st.w_size:
          hip.momentum_flat_dev(st.w_master, w_grad, self.slots_w[0], self.w_mask, st.w_decay, wd, self.g_scale, self.hp,
                                self.momentum)
        if st.o_size:
          hip.momentum_flat_dev(st.o_master, o_grad, self.slots_o[0], self.o_mask, st.o_decay, wd, self.g_scale, self.hp,
                                self.momentum)
      st.w_t_fresh = False
      st.zero_grad()
      return
    if self.kind == 'adam':
      b1p, b2p = float(self.beta1_power), float(self.beta2_power)
      if st.w_size:
        hip.adam_flat(st.w_master, w_grad, self.slots_w[0], self.slots_w[1], self.w_mask, st.w_decay, wd,
                      self.g_scale, lrn_rate, self.beta1, self.beta2, self.epsilon, b1p, b2p)
      if st.o_size:
        hip.adam_flat(st.o_master, o_grad, self.slots_o[0], self.slots_o[1], self.o_mask, st.o_decay, wd,
                      self.g_scale, lrn_rate, self.beta1, self.beta2, self.epsilon, b1p, b2p)
      self.beta1_power = np.float32(self.beta1_power * np.float32(self.beta1))
      self.beta2_power = np.float32(self.beta2_power * np.float32(self.beta2))
    else:
      if st.w_size:
        hip.momentum_flat(st.w_master, w_grad, self.slots_w[0], self.w_mask, st.w_decay, wd, self.g_scale,
                          lrn_rate, self.momentum)
      if st.o_size:
        hip.momentum_flat(st.o_master, o_grad, self.slots_o[0], self.o_mask, st.o_decay, wd, self.g_scale,
                          lrn_rate, self.momentum)
    st.w_t_fresh = False          # the backward-data layout of the kernels is stale when w_master aliases w_compute
    st.zero_grad()


class GradReducer(object):
  """Gradient exchange of one VarStore: all-reduce(sum) of its flat gradient buffers over RCCL, launched from inside
  the backward pass so that it overlaps with the remaining backward kernels (Horovod's fused, in-backward all-reduce,
  reference utils/multi_gpu_wrapper.py:83-90).

  * The matmul-kernel gradients (`w_grad`, 23.4 M elements for ResNet-50) are cut into contiguous buckets of
    ~`bucket_elems` elements in buffer order.  The backward pass produces gradients from the END of the buffer towards
    its beginning (layers are laid out in creation order); each gradient producer calls `VarStore.notify_grad(var)`
    (fused convolutions write the flat buffer directly; every other leaf reports through a post-accumulate hook), and
    when the last variable of a bucket has reported, that bucket's slice is all-reduced asynchronously.  xGMI is a
    point-to-point mesh (7 links x ~153 GB/s per GPU): few, large messages keep every link busy, so buckets are tens of
    MB, not per-tensor messages.
  * Reduction dtype.  In bf16 compute mode the gradient buffer is bf16; summing 8 bf16 buffers in bf16 loses up to
    ~3 bits (tests/test_multi_gpu_gloo.py bounds it), so by default every bucket is first widened into a float32
    staging buffer and reduced there (`--allreduce_dtype float32`); the optimiser kernel then reads the staging
    buffer.  `--allreduce_dtype compute` reduces in the buffer's own dtype (half the bytes on the links).
  * `finish()` (called by DistributedFlatOptimizer.compute_gradients) launches whatever has not been launched
    (variables that received no gradient this step), reduces the small `o_grad` buffer, waits for all handles and
    returns the tensors the optimiser must read.  A variable reporting twice in one cycle (a second backward pass
    before the update) invalidates the overlapped launches: everything is re-staged from the intact source buffers
    and reduced again, blocking -- always correct, only slower.
  """

  def __init__(self, store: VarStore, bucket_elems: int = 8 << 20, reduce_dtype: Optional[torch.dtype] = torch.float32,
               overlap: bool = True):
    self.store = store
    self.overlap = overlap
    self.reduce_dtype = reduce_dtype
    self.buckets = []          # [lo, hi, n_vars]
    self.var_bucket = {}
    self._build_buckets(bucket_elems)
    self._stage_w: Optional[torch.Tensor] = None
    self._stage_o: Optional[torch.Tensor] = None
    self._reset_cycle()
    self.armed = False         # in-backward launching is OPT-IN per step: arm() ... backward ... disarm() / finish()
    store.grad_hook = self._on_grad
    self.n_overlapped = 0      # buckets launched from inside backward in the last cycle (diagnostics / tests)
    # While the learner's step is recorded in hipGraphs (step_graph.py): the backend whose `cut(action)` takes the exchange -- the
    # staging copies are captured, the all-reduces stay host calls BETWEEN the two graphs of the chain, made again in every
    # replay.  None (and outside a capture): collectives are issued on the spot.
    self.recorder = None

  def _build_buckets(self, bucket_elems: int) -> None:
    ws = sorted([v for v in self.store.vars if v.group == 'W' and v.trainable], key=lambda v: v.offset)
    lo, cnt = 0, 0
    for i, v in enumerate(ws):
      self.var_bucket[v.name] = len(self.buckets)
      cnt += 1
      end = v.offset + v.numel
      last = i == len(ws) - 1
      if last:
        end = self.store.w_size
      if end - lo >= bucket_elems or last:
        self.buckets.append([lo, end, cnt])
        lo, cnt = end, 0

  def _reset_cycle(self) -> None:
    self.pending = [b[2] for b in self.buckets]
    self.seen = set()
    self.launched = [False] * len(self.buckets)
    self.handles = []
    self.dirty = False

  def _active(self) -> bool:
    return dist.is_initialized() and dist.get_world_size() > 1

  def _stage(self, which: str) -> torch.Tensor:
    st = self.store
    src = st.w_grad if which == 'w' else st.o_grad
    dt = self.reduce_dtype or src.dtype
    buf = self._stage_w if which == 'w' else self._stage_o
    if buf is None or buf.dtype != dt:
      buf = torch.empty(src.numel(), dtype=dt, device=src.device)
      if which == 'w':
        self._stage_w = buf
      else:
        self._stage_o = buf
    return buf

  def _stage_bucket(self, b: int):
    """Widen bucket b into the staging buffer (stream-ordered copy); -> the call that all-reduces it."""
    lo, hi, _ = self.buckets[b]
    piece = self._stage('w')[lo:hi]
    side = getattr(self.store, 'wrw_side', None)
    if side is not None and side.forks and self.store.w_grad.is_cuda:
      # graph.WrwSide: backward-filter launches of this pass run on a second stream.  A bucket completed by one of them is staged on
      # that stream (its notification is issued there, and the stream waited for the main stream when it forked); a bucket completed
      # from the main stream may hold gradients whose launches are still running over there
      cur = torch.cuda.current_stream(self.store.w_grad.device)
      if cur != side.stream:
        cur.wait_stream(side.stream)
    piece.copy_(self.store.w_grad[lo:hi])
    self.launched[b] = True
    return lambda: self.handles.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True))

  def _launch(self, b: int) -> None:
    self._stage_bucket(b)()

  def _recording(self) -> bool:
    """The step is being captured into hipGraphs right now (step_graph.CudaBackend): nothing may go out from inside the backward
    pass -- a capture ended and re-begun from a gradient hook replayed garbage in one bucket from the second replay on (measured,
    profiles/r05_recorded_step_two_ranks.txt) -- so the whole exchange of a recorded step is made in finish(), between TWO graphs."""
    return self.recorder is not None and getattr(self.recorder, 'capturing', False)

  def arm(self) -> None:
    """Allow bucket all-reduces to be launched from inside the NEXT backward pass.  Every rank must run that backward
    (it is the one followed by finish()); a backward that only some ranks run -- LayerwiseTuner on the primary worker
    while the others sit in a barrier, regression-gradient helpers that do their own all-reduce -- must stay unarmed,
    or its collectives would pair with the other ranks' barrier.  Arming starts a fresh cycle: whatever an unarmed or
    abandoned backward left behind (seen-set, pending counts) is dropped."""
    for h in self.handles:       # an armed backward that never reached finish(): its launches are complete collectives
      h.wait()
    self._reset_cycle()
    self.armed = True

  def disarm(self) -> None:
    self.armed = False

  def _on_grad(self, var) -> None:
    if not (self.overlap and self._active()) or var.group != 'W' or self._recording():
      return
    if not self.armed:
      # a gradient produced OUTSIDE the armed pass while buckets of this cycle are already in flight (a second,
      # accumulating backward before the update): the overlapped results are stale -> finish() re-reduces, blocking
      if any(self.launched):
        self.dirty = True
      return
    b = self.var_bucket.get(var.name)
    if b is None:
      return
    if var.name in self.seen:
      # Inside ONE armed backward pass a variable reports twice when its producer writes the flat buffer directly: once
      # through VarStore.notify_grad from the kernel launcher, once through the leaf's post-accumulate hook, which torch
      # (2.10) fires even when the autograd function returned None for that leaf.  Both come after the gradient was
      # enqueued, so the second is a no-op.  (Round 2 marked the cycle dirty here: on the fused bf16 path EVERY kernel
      # reported twice, every overlapped launch was thrown away and re-done blocking in finish() -- found by the
      # `buckets_launched_inside_backward > 0` assertion of tests/test_learner_gpu.py, round 3.)  A second, accumulating
      # backward pass is outside the armed window and still invalidates the launches (branch above).  Limitation: a kernel
      # shared by two ops of one graph would report after its FIRST consumer; no network of the path shares kernels.
      return
    self.seen.add(var.name)
    self.pending[b] -= 1
    if self.pending[b] == 0 and not self.launched[b]:
      self._launch(b)
      self.n_overlapped += 1

  def finish(self):
    """-> (w_grad tensor, o_grad tensor, g_scale) the optimiser kernel must use for this update."""
    st = self.store
    self.armed = False
    if not self._active():
      self._reset_cycle()
      return st.w_grad, st.o_grad, 1.0
    n_over = sum(self.launched)
    if self.dirty:
      for h in self.handles:
        h.wait()
      self.handles = []
      self.launched = [False] * len(self.buckets)
      n_over = 0
    calls = []
    if st.w_size:
      calls = [self._stage_bucket(b) for b in range(len(self.buckets)) if not self.launched[b]]
    w_src = self._stage('w') if st.w_size else st.w_grad
    o_src = st.o_grad
    if st.o_size:
      o_src = self._stage('o')
      o_src.copy_(st.o_grad)
    o_piece = o_src if st.o_size else None

    def exchange():            # what is left of the step's exchange: the buckets not launched from inside backward, the small buffer;
      for call in calls:       # then every collective of the step is waited for
        call()
      if o_piece is not None:
        self.handles.append(dist.all_reduce(o_piece, op=dist.ReduceOp.SUM, async_op=True))
      for h in self.handles:
        h.wait()
      self.handles = []
    if self.recorder is not None:
      self.recorder.cut(exchange)       # recorded step: a host call between the two graphs of the chain, made again in every replay
    else:
      exchange()
    self._reset_cycle()
    self.n_overlapped = n_over
    return w_src, o_src, 1.0 / dist.get_world_size()


def grad_reducer_of(store: VarStore) -> GradReducer:
  """The store's (single) reducer: several optimisers may wrap one store (regression + fine-tune optimisers)."""
  r = getattr(store, 'reducer', None)
  if r is None:
    import os
    from pocketflow_amd.flags import FLAGS
    rd = {'float32': torch.float32, 'compute': None}[FLAGS.allreduce_dtype if 'allreduce_dtype' in FLAGS else 'float32']
    r = GradReducer(store, bucket_elems=int(os.environ.get('PF_ALLREDUCE_BUCKET', 8 << 20)), reduce_dtype=rd,
                    overlap=os.environ.get('PF_OVERLAP_ALLREDUCE', '1') != '0')
    store.reducer = r
  return r


class DistributedFlatOptimizer(object):
  """mgw.DistributedOptimizer(optimizer): all-reduce(sum) of the flat gradient buffers (GradReducer: bucketed,
  launched from inside backward), then the wrapped optimiser with g_scale = 1 / world_size (Horovod's average).
  Masks are applied AFTER the reduction, as in the reference (ws learner.py:205-207).

  Attribute reads AND writes are forwarded to the wrapped optimiser: learners set `optimizer.weight_decay`,
  `.w_mask`, `.o_mask` on whatever `mgw.DistributedOptimizer` returned, and apply_gradients() of the wrapped
  object must see them (Horovod's wrapper subclasses the optimiser, so attributes are shared there too)."""

  _OWN = ('opt', 'reducer')

  def __init__(self, optimizer: FlatOptimizer):
    object.__setattr__(self, 'opt', optimizer)
    object.__setattr__(self, 'reducer', grad_reducer_of(optimizer.store))

  def __getattr__(self, name):
    return getattr(self.opt, name)

  def __setattr__(self, name, value):
    if name in DistributedFlatOptimizer._OWN:
      object.__setattr__(self, name, value)
    else:
      setattr(self.opt, name, value)

  def backward(self, loss: torch.Tensor) -> None:
    """loss.backward() with the in-backward bucket launches of the GradReducer enabled for exactly this pass."""
    from pocketflow_amd import graph as G
    self.reducer.arm()
    try:
      with G.wrw_side_armed(self.opt.store):
        loss.backward()
    finally:
      self.reducer.disarm()

  def compute_gradients(self) -> None:
    w_src, o_src, scale = self.reducer.finish()
    self.opt.w_grad_src, self.opt.o_grad_src = w_src, o_src
    self.opt.g_scale = scale

  def apply_gradients(self, lrn_rate: float) -> None:
    self.opt.apply_gradients(lrn_rate)
