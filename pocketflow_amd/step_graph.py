"""The steady-state fine-tune step recorded ONCE in a hipGraph and replayed (`--enbl_step_graph`).

Why.  A step of this engine is 300-700 kernel launches issued from Python: 10 ms of host time for ResNet-50 (the GPU needs 24 ms:
fine), 4.4 ms for ResNet-20 @ CIFAR-10 (the GPU needs less: the step is HOST-bound, profiles/r03_host_overhead_c1.txt) and 15 ms for
MobileNet-v1.  The reference has the same structure one level up: `sess.run(train_op)` replays a graph TensorFlow compiled once
(learners/uniform_quantization/learner.py:172-189), its learning rate and bit widths are tensors of that graph.  Here the shapes are
static, every parameter / gradient / statistic lives in flat buffers that never move, and the kernels take a stream argument -- so
the whole step (weight fake-quant, forward, losses, backward with straight-through estimators, fused optimiser update) can be
captured from the eager code as it is and replayed with ONE host call.

What changes per step and therefore cannot be a by-value kernel argument:
  * the batch                       -> static input buffers, filled before the replay (device-to-device copies in stream order);
  * the learning rate, Adam's bias correction
                                    -> `FlatOptimizer.enable_device_hyper()`: the update kernels read them from device memory
                                       (pf_adam_flat_dev / pf_momentum_flat_dev), `feed_hyper` writes them before the replay;
  * anything a network draws per step on the host (MobileNet's dropout mask)
                                    -> `graph.step_feeders`: callables run before every replay (stream-ordered copies into the
                                       buffer the captured launch reads).
The frozen teacher's forward pass over the NEXT batch (learners/teacher_ahead.py) is part of the graph as a forked branch: it
runs on a second stream from the start of the step (beside the student's forward pass: that is where it pays, measured) and joins
before the tail, where `next` becomes `current` (three device copies).

Eager steps and graph steps can alternate (`suspend()` / `resume()`): both draw from the same iterator in the same order and hand
the prefetched batch over through `TeacherAhead.pending`.  Whatever goes wrong while recording (an op that synchronises, a library
call that cannot be captured) is reported ONCE and the learner stays on the eager path -- the result of a step never depends on
the mode (tests/test_learner_gpu.py compares them bit for bit).
"""
from __future__ import annotations

import contextlib
import logging
import os

import torch

from pocketflow_amd import profiling
from pocketflow_amd.flags import FLAGS
from pocketflow_amd.learners import teacher_ahead

log = logging.getLogger('pocketflow_amd')


class CudaBackend(object):
  """torch.cuda.CUDAGraph is a hipGraph on ROCm.

  One process: the step is ONE graph.  Several ranks (`segmented`): the step is TWO graphs around the gradient exchange --
  `cut(action)`, called by optim.GradReducer.finish() behind the recorded backward pass and the captured staging copies of the
  gradient buckets, ends the graph under capture, notes `action` (the all-reduces of the step: ordinary host-side RCCL calls,
  then the wait for them) and starts the next graph (the optimiser update and the tail) in the same memory pool; a replay is
  graph 0, action, graph 1 -- three host calls plus a handful of collectives instead of ~700 launches (Horovod's all-reduce
  lives inside the reference's compiled train graph the same way: utils/multi_gpu_wrapper.py:83-98,
  learners/uniform_quantization/learner.py:246).  The collectives themselves are NOT captured: RCCL runs them on its own stream,
  ordered behind graph 0 by the usual event hand-shake of an asynchronous collective, so nothing here depends on graph support
  inside RCCL.  What a recorded step gives up is the overlap of the exchange with the rest of the backward pass (the
  launch-by-launch step keeps it): cutting from the gradient hooks INSIDE the captured backward pass was built first and
  replayed garbage in one bucket from the second replay on (profiles/r05_recorded_step_two_ranks.txt); with the cut behind
  backward the recorded and the launch-by-launch run agree.  Segmented captures run in thread-local capture mode (the process
  group's watchdog thread polls events of earlier collectives; a global-mode capture would be invalidated by it) with the backward
  pass on the calling thread, as do the launch-by-launch steps before them (`warm`)."""

  def __init__(self, device, segmented=False):
    self.device = device
    self.segmented = segmented
    self.graph = torch.cuda.CUDAGraph()
    self.graphs = [self.graph]              # the step in replay order: graphs[0], actions[0], graphs[1], ...
    self.actions = []
    self.side = torch.cuda.Stream(device=device)
    self._joined = True
    self.capturing = False

  def capture(self, body):
    torch.cuda.synchronize(self.device)
    if not self.segmented:
      with torch.cuda.graph(self.graph):
        out = body(self)
    else:
      out = self._capture_segments(body)
    torch.cuda.synchronize(self.device)
    return out

  def _capture_segments(self, body):
    torch.cuda.empty_cache()
    stream = torch.cuda.Stream(device=self.device)
    with torch.cuda.stream(stream), torch.autograd.set_multithreading_enabled(False):
      self.graph.capture_begin(capture_error_mode='thread_local')
      self.capturing = True
      try:
        out = body(self)
      except BaseException:
        self.capturing = False
        try:
          self.graphs[-1].capture_end()
        except Exception:        # pylint: disable=broad-except
          pass                    # (the capture is already invalid: the first error is the one to report)
        raise
      self.capturing = False
      self.graphs[-1].capture_end()
    return out

  def warm(self):
    """Context of the launch-by-launch steps that precede a segmented recording: their backward pass runs on the calling thread
    too.  aten's per-thread library handles (hipBLASLt creates one on a thread's first matrix product, which allocates) must
    exist on the thread that will capture: created inside the capture they end it with hipErrorStreamCaptureUnsupported."""
    return torch.autograd.set_multithreading_enabled(False) if self.segmented else contextlib.nullcontext()

  def cut(self, action):
    """End the graph under capture here; `action()` runs on the host between the two graphs in every replay.  Outside a capture
    (launch-by-launch steps of a learner whose step is also recorded) the action simply runs."""
    if not self.capturing:
      action()
      return
    self.join()                             # a forked branch must have joined before a capture ends
    self.graphs[-1].capture_end()
    self.actions.append(action)
    g = torch.cuda.CUDAGraph()
    g.capture_begin(pool=self.graph.pool(), capture_error_mode='thread_local')
    self.graphs.append(g)

  @contextlib.contextmanager
  def fork(self):
    """A branch of the graph: the side stream joins the capture by waiting for the capturing stream."""
    self.side.wait_stream(torch.cuda.current_stream(self.device))
    self._joined = False
    with torch.cuda.stream(self.side):
      yield

  def join(self):
    if not self._joined:
      torch.cuda.current_stream(self.device).wait_stream(self.side)
      self._joined = True

  def replay(self):
    for i, g in enumerate(self.graphs):
      g.replay()
      if i < len(self.actions):
        self.actions[i]()
    return None

  def recover(self):
    torch.cuda.synchronize(self.device)


class InlineBackend(object):
  """No graph (CPU emulation of the kernels in tests/): "recording" keeps the body, a "replay" executes it -- the control flow,
  the static buffers, the feeders and the hand-over between the modes are the ones of the real backend."""

  def __init__(self):
    self.body = None

  def capture(self, body):
    self.body = body
    return None

  def fork(self):
    return contextlib.nullcontext()

  def join(self):
    pass

  def warm(self):
    return contextlib.nullcontext()

  def cut(self, action):
    action()

  def replay(self):
    return self.body(self)

  def recover(self):
    pass


def _step_attr(learner) -> str:
  return 'ft_step' if hasattr(learner, 'ft_step') else 'global_step'


def _with_lr(out, lr):
  """The step's return value with this step's learning rate in it (UQ / NUQ: a dict with 'lr'; WS / CP: (lr, loss, metrics))."""
  if isinstance(out, dict):
    r = dict(out)
    r['lr'] = lr
    return r
  return (lr,) + tuple(out[1:])


class StepGraph(object):
  WARM = 3          # eager steps before recording: allocator pools, MIOpen / rocBLAS handles and solver choices, launch attributes

  def __init__(self, learner, backend):
    self.learner, self.backend = learner, backend
    self.state = 'warm'                     # warm -> ready | failed
    self.n_eager = self.n_replays = 0
    self.suspended = False
    self.out = None
    self.cur = self.nxt = None              # static (x, y, teacher logits) of the step / of the next step (distillation only)
    self.nxt_raw = None                     # the iterator's (images, labels) behind `nxt` (handed back when the graph is suspended)
    self.nxt_stale = False                  # `nxt` was consumed by a replay: draw the following batch before the next one
    self.cur_raw = None                     # without a teacher: the batch already loaded into `cur` (not yet consumed by a replay)
    self.cur_images = None                  # with a teacher: the images behind `cur` as the iterator delivered them (teacher_ahead.next_images)
    self.error = None
    self.lookahead_void = False             # the iterator was reset behind the batches held here: they belong to its previous pass
    self.auto_suspended = False             # suspended for ONE foreign consumer of the iterator; the next plain step resumes

  # -- mode switches --------------------------------------------------------------------------------
  def suspend(self):
    """Following steps run eagerly (bench.py times a few steps kernel by kernel with HIP events; a search changes bit widths)."""
    if self.state == 'ready' and not self.suspended:
      self._hand_to_eager()
    if self.state == 'ready':
      self.learner.optimizer.hyper_external = False
    self.suspended = True
    self.auto_suspended = False

  def resume(self):
    if self.suspended and self.state == 'ready':
      self.learner.optimizer.hyper_external = True
      self._load_current()
    self.suspended = False
    self.auto_suspended = False

  def yield_to_eager(self):
    """Another consumer of the training iterator runs NOW (a `train_step` with arguments, layer-wise tuning's `next_images`): hand
    the batches held in the static buffers over, so that it sees the batch the next replay would have seen; the next plain step
    resumes the replays by itself (those consumers used to read PAST the held batches)."""
    if self.state == 'ready' and not self.suspended:
      self.suspend()
      self.auto_suspended = True

  def discard_lookahead(self):
    """The training iterator was reset or re-built: the batches held here were drawn from its previous pass.  Forget them WITHOUT
    handing them back (teacher_ahead.drop calls this); the next replay re-fills the static buffers from the iterator."""
    self.nxt_raw = self.cur_raw = self.cur_images = None
    if self.state == 'ready' and not self.suspended:
      self.lookahead_void = True

  def invalidate(self):
    """Something the recorded launches carry by value changed (bit widths, masks re-built, optimiser replaced): record again."""
    if self.state == 'ready':
      self.suspend()
      self.learner.optimizer.hyper_external = False
    red = _reducer_of(self.learner)
    if red is not None:
      red.recorder = None                    # (the old backend must not outlive its graphs inside the store-wide reducer: ADVICE r5)
    if self.state != 'failed':
      self.state, self.n_eager, self.suspended = 'warm', 0, False
      self.out = self.cur = self.nxt = self.nxt_raw = self.cur_raw = self.cur_images = None
      self.nxt_stale = self.lookahead_void = self.auto_suspended = False
      self.backend = _new_backend(self.learner)

  # -- one step ---------------------------------------------------------------------------------------
  def step(self):
    lrn = self.learner
    if self.suspended and self.auto_suspended and self.state == 'ready':
      self.resume()
    if self.state == 'failed' or self.suspended:
      return lrn._train_step_eager()
    if self.state == 'warm':
      if self.n_eager < self.WARM:
        self.n_eager += 1
        with self.backend.warm():
          return lrn._train_step_eager()
      try:
        self._record()
      except Exception as e:        # pylint: disable=broad-except
        self.error = e
        self.state = 'failed'
        lrn._static_batch = None
        lrn.optimizer.hyper_external = False
        lrn.graph.capturing = False
        red = _reducer_of(lrn)
        if red is not None:
          red.recorder = None
        try:
          self.backend.recover()
          if self.cur is not None:
            self._hand_to_eager()                          # the batches drawn for the static buffers stay first in line
        except Exception as e2:     # pylint: disable=broad-except
          log.warning('step graph: clean-up after the failed recording: %s', e2)
        log.warning('step graph: recording failed (%s: %s) -- the learner stays on the eager path', type(e).__name__, e)
        if os.environ.get('PF_STEP_GRAPH_STRICT') and _reducer_of(lrn) is None:
          raise
      if not self._ranks_agree():
        return lrn._train_step_eager()
    return self._replay()

  def _ranks_agree(self) -> bool:
    """Several ranks: the job stays on the recorded step only if EVERY rank recorded it.  A rank that fell back alone
    would send its bucket all-reduces from the backward hooks in completion order while the others issue buckets 0..n-1 from
    finish() between their two graphs -- collectives of different order and size across ranks.  Every rank reaches this point in
    the same step (WARM eager steps, then the recording attempt), so the MIN all-reduce of the success flag pairs up.  Returns
    whether this rank replays."""
    lrn = self.learner
    ok = self.state == 'ready'
    if _reducer_of(lrn) is None:
      return ok
    import torch.distributed as dist
    flag = torch.tensor([1.0 if ok else 0.0], device=lrn.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if float(flag.item()) >= 1.0:
      return True
    if ok:                                                   # another rank could not record: back to launch-by-launch steps with it
      log.warning('step graph: another rank could not record its step -- every rank stays on the eager path')
      self.suspend()
      self.state = 'failed'
      red = _reducer_of(lrn)
      if red is not None:
        red.recorder = None
    elif os.environ.get('PF_STEP_GRAPH_STRICT') and self.error is not None:
      raise self.error
    return False

  # -- internals ---------------------------------------------------------------------------------------
  def _fetch(self, keep_raw=False):
    lrn = self.learner
    raw = teacher_ahead.fetch_raw(lrn)
    if keep_raw:
      self.nxt_raw = raw
    return lrn.to_device(*raw)

  def _load_current(self):
    """Fill the static buffers of the NEXT replay: distillation -> (current from what an eager step prefetched or from the iterator
    with the teacher in line, next from the iterator); otherwise nothing (the batch is fetched right before each replay)."""
    lrn = self.learner
    self.lookahead_void = False
    if self.nxt is None:
      return
    ahead = teacher_ahead.of(lrn)
    if ahead is not None and ahead.pending is not None:
      self.cur_images = ahead.pending[4]
      x, y, logits = ahead.take()
    else:
      raw = teacher_ahead.fetch_raw(lrn)
      self.cur_images = raw[0]
      x, y = lrn.to_device(*raw)
      logits = teacher_ahead.teacher_of(lrn).calc_logits(None, x)
    for dst, src in zip(self.cur, (x, y, logits)):
      dst.copy_(src)
    x, y = self._fetch(keep_raw=True)
    self.nxt[0].copy_(x)
    self.nxt[1].copy_(y)
    self.nxt_stale = False

  def _hand_to_eager(self):
    """The batch in `current` (its teacher logits are computed) is the next one in data order: the eager path takes it first."""
    lrn = self.learner
    if self.lookahead_void:                                # (drawn from a pass of the iterator that is over: nothing to hand over)
      self.lookahead_void = False
      self.nxt_raw = self.cur_raw = None
      return
    if self.nxt is None:
      if self.cur_raw is not None:                         # drawn, never consumed: back in front of the iterator
        lrn.__dict__.setdefault('_unget', []).insert(0, self.cur_raw)
        self.cur_raw = None
      return
    x, y, logits = (t.clone() for t in self.cur)
    ahead = teacher_ahead.of(lrn)
    if ahead is None:                                      # (PF_TEACHER_AHEAD=0 with distillation: a helper just for the hand-over --
      ahead = lrn._teacher_ahead = teacher_ahead.TeacherAhead(lrn, teacher_ahead.InlineStreams())   # gone again once the batch is taken)
      ahead.handover_only = True
    if ahead.pending is not None:
      ahead.drop()
    ahead.pending = (x, y, logits, None, self.cur_images)
    # The eager helper's side stream must not run ahead of the replays that are still queued: its next teacher forward would run
    # CONCURRENTLY with the teacher branch of a replay that has not executed yet -- two forward passes of one frozen network through
    # the same scratch buffers (seen as a loss mismatch in tests/step_graph_worker.py when the side stream was left free).
    ahead.streams.side_waits_for_main()
    # a `next` that no replay has consumed yet holds one more batch drawn from the iterator: back it goes, in front of the iterator
    if not self.nxt_stale and self.nxt_raw is not None:
      lrn.__dict__.setdefault('_unget', []).insert(0, self.nxt_raw)
    self.nxt_raw = None

  def _record(self):
    lrn = self.learner
    opt = lrn.optimizer
    g = lrn.graph
    teacher = teacher_ahead.teacher_of(lrn) if FLAGS.enbl_dst else None
    # static inputs
    ahead = teacher_ahead.of(lrn) if teacher is not None else None
    if ahead is not None and ahead.pending is not None:
      self.cur_images = ahead.pending[4]
      x, y, logits = ahead.take()
    elif teacher is not None:
      raw = teacher_ahead.fetch_raw(lrn)
      self.cur_images = raw[0]
      x, y = lrn.to_device(*raw)
      logits = teacher.calc_logits(None, x)
    else:
      self.cur_raw = teacher_ahead.fetch_raw(lrn)            # no look-ahead without a teacher: this batch is the first replay's
      x, y = lrn.to_device(*self.cur_raw)
      logits = None
    self.cur = (x.clone(), y.clone(), logits.clone() if logits is not None else None)
    if teacher is not None:
      nx, ny = self._fetch(keep_raw=True)
      self.nxt = (nx.clone(), ny.clone(), torch.empty_like(self.cur[2]))
    opt.enable_device_hyper()
    opt.hyper_external = True
    g.capturing = True
    cur, nxt = self.cur, self.nxt
    red = _reducer_of(lrn)
    if red is not None:
      red.recorder = self.backend                            # the exchange goes out through backend.cut(): between the two graphs

    def body(be):
      if nxt is not None:
        # the teacher over `next` from the START of the step, beside the forward pass (forked at the start of the BACKWARD pass
        # instead: -0.5 %, profiles/r06_policy_ab.txt)
        with be.fork(), profiling.suspended():
          nxt[2].copy_(teacher.calc_logits(None, nxt[0]))
      lrn._static_batch = cur
      try:
        out = lrn._train_step_eager()
      finally:
        lrn._static_batch = None
      if nxt is not None:
        be.join()
        for dst, src in zip(cur, nxt):
          dst.copy_(src)
      return out
    step0 = getattr(lrn, _step_attr(lrn))
    pow0 = (opt.beta1_power, opt.beta2_power)
    try:
      self.out = self.backend.capture(body)
      if self.out is not None:
        from pocketflow_amd.learners.abstract_learner import _detached
        self.out = _detached(self.out)                       # static output tensors; the recorded step's Python graph is not needed
    except BaseException:
      if red is not None:
        red.recorder = None
      raise
    finally:
      g.capturing = False
      # recording executed nothing on the device; undo the host-side bookkeeping of the recorded step (also when the recording failed
      # half-way: the eager path that takes over must continue from the step counter / Adam powers of the last EXECUTED step)
      setattr(lrn, _step_attr(lrn), step0)
      opt.beta1_power, opt.beta2_power = pow0
    self.state = 'ready'
    log.info('step graph: recorded the %s step (%s%s)', type(lrn).__name__, 'teacher forked on a side stream' if nxt is not None else 'single stream',
             '; %d graphs around %d gradient-exchange calls' % (len(self.backend.graphs), len(self.backend.actions))
             if getattr(self.backend, 'actions', None) else '')

  def _replay(self):
    lrn = self.learner
    attr = _step_attr(lrn)
    step = getattr(lrn, attr)
    lr = lrn.lrn_rate(step)
    if self.lookahead_void:                                # the iterator was reset: both static batches are re-drawn from it
      self.cur_raw = None
      self._load_current()
    if self.nxt is None:
      if self.cur_raw is not None:
        self.cur_raw = None                                  # loaded when the step was recorded / resumed
      else:
        x, y = self._fetch()
        self.cur[0].copy_(x)
        self.cur[1].copy_(y)
    elif self.nxt_stale:
      x, y = self._fetch(keep_raw=True)
      self.nxt[0].copy_(x)
      self.nxt[1].copy_(y)
    lrn.optimizer.feed_hyper(lr)
    for feed in getattr(lrn.graph, 'step_feeders', []):
      feed()
    out = self.backend.replay()
    if out is not None:                                    # in-line stand-in: the body ran and advanced the counter itself
      self.out = out
    else:
      setattr(lrn, attr, step + 1)
    self.n_replays += 1
    self.nxt_stale = True                                  # the tail of the replay moved `next` into `current`
    if self.nxt_raw is not None:
      self.cur_images = self.nxt_raw[0]
    return _with_lr(self.out, lr)


def _reducer_of(learner):
  """The learner's gradient exchange when there is one to make (several ranks): optim.GradReducer, else None."""
  red = getattr(learner.optimizer, 'reducer', None)
  return red if red is not None and red._active() else None       # pylint: disable=protected-access


def _new_backend(learner):
  if os.environ.get('PF_STEP_GRAPH') == 'inline':
    return InlineBackend()
  if torch.device(learner.device).type == 'cuda':
    return CudaBackend(learner.device, segmented=_reducer_of(learner) is not None)
  return None


def of(learner) -> StepGraph:
  sg = getattr(learner, '_step_graph', None)
  if sg is None:
    backend = _new_backend(learner)
    sg = learner._step_graph = StepGraph(learner, backend)
    if backend is None:
      sg.state = 'failed'                                  # no HIP device: eager
  return sg


def invalidate(learner) -> None:
  sg = getattr(learner, '_step_graph', None)
  if sg is not None:
    sg.invalidate()
