"""Teacher forward of the NEXT batch on a second HIP stream (default for the distillation learners since round 4; PF_TEACHER_AHEAD=0
switches it off).  Measured in one box (profiles/r04_first_call_ab.txt): ResNet-50 UQ w8/a8 + distillation at B = 256, 9 735 -> 10 379
images/s; NUQ 4-bit 9 716 -> 10 382; the distillation / bf16 parity tests pass with it.

The distillation teacher is frozen: its logits for a batch depend on nothing the fine-tune step changes
(learners/distillation_helper.py:62-64 of the reference builds it under `tf.stop_gradient`).  The step therefore does not have to
run teacher -> student -> backward in one queue: with this helper the learner issues, at the END of step k, the upload of batch k+1
and the teacher's forward pass over it on a side stream, and step k+1 starts from those logits.  Same work per step, same values
(the teacher is deterministic) -- but the teacher's launches execute while the main stream is still working through step k's
backward pass.  Why that can pay on this design: every contraction kernel is a persistent launch whose workgroups own
ceil(tiles / slots) tiles (1.53 -> 2, 3.06 -> 4 rounds on the 14x14 / 7x7 layers); in ONE queue the CUs that
finished their last tile idle until the slowest workgroup is done, with a second queue the next kernel's workgroups take them.

Stream discipline:
* everything of batch k+1 (upload, layout / dtype conversion, teacher forward) is issued on the side stream and allocated from its
  pool; the tensors the main stream will read (x, y, logits) are `record_stream`ed for it, so the caching allocator does not hand
  their memory to a later side-stream allocation while main is still reading them;
* main waits for ONE event recorded behind the teacher forward before it touches them;
* the side stream never waits for main: the teacher's weights and scratch were written long before (first, in-line teacher
  forward of step 0), the batch comes from the host.
The roofline region of bench.py is suspended while the teacher is issued (its launches run beside backward kernels: their
event-to-event durations are not a kernel's duration any more); the student's launches of the same kernels remain in it.
"""
from __future__ import annotations

import contextlib
import os

import torch

from pocketflow_amd import profiling


def enabled() -> bool:
  return os.environ.get('PF_TEACHER_AHEAD', '1') not in ('', '0')


class CudaStreams(object):
  """The HIP side of the helper (torch.cuda is HIP on ROCm)."""

  def __init__(self, device):
    self.device = device
    self.side = torch.cuda.Stream(device=device)

  def on_side(self):
    return torch.cuda.stream(self.side)

  def side_waits_for_main(self):
    """What is issued on the side stream next starts when the main stream has reached THIS point."""
    self.side.wait_stream(torch.cuda.current_stream(self.device))

  def record(self):
    ev = torch.cuda.Event()
    ev.record(self.side)
    return ev

  def main_waits(self, ev):
    torch.cuda.current_stream(self.device).wait_event(ev)

  def hand_to_main(self, t):
    if t is not None and t.is_cuda:
      t.record_stream(torch.cuda.current_stream(self.device))


class InlineStreams(object):
  """No second queue (CPU emulation of the kernels in tests/): the same control flow, executed in program order."""

  def on_side(self):
    return contextlib.nullcontext()

  def side_waits_for_main(self):
    pass

  def record(self):
    return None

  def main_waits(self, ev):
    pass

  def hand_to_main(self, t):
    pass


def fetch_raw(learner):
  """The next (images, labels) of the training data in iterator order: batches that were drawn but not consumed (a suspended
  step graph hands its look-ahead batch back) come first."""
  back = getattr(learner, '_unget', None)
  if back:
    return back.pop(0)
  return learner.iter_train.get_next()


class TeacherAhead(object):
  def __init__(self, learner, streams):
    self.learner, self.streams, self.pending = learner, streams, None
    self.n_issued = self.n_taken = 0
    self.handover_only = False    # made by a suspended step graph for ONE hand-over although the user switched the helper off
    self.serialise = False        # bench.py (roofline.unshared): the side stream waits for the main stream before the teacher is issued

  def issue(self):
    """Fetch batch k+1, upload it and run the teacher over it -- all on the side stream."""
    lrn, st = self.learner, self.streams
    teacher = teacher_of(lrn)
    # Called at the END of step k; the side stream does NOT wait for the main stream: the launches start as soon as they are issued and
    # slip into whatever the main stream is executing then (the host is about a step ahead of the GPU).  Round 4 measured the
    # alternatives (profiles/r04_overlap_ab.txt): this, +6.6 % (9 735 -> 10 379); side stream held until step k is done, so that the
    # teacher runs beside step k + 1's forward pass as it does inside a RECORDED step (where the two branches genuinely time-share the
    # chip and it is worth +6 %): -6 % launch by launch (9 251 vs 9 839 recorded in one box -- launch by launch the persistent kernels
    # of the two streams alternate instead of sharing); teacher beside the backward pass +0.5 %; backward-filter launches on a third
    # stream -2.8 % THEN (round 4's filter kernels filled the chip for 100-250 us each; with round 6's kernels the same fork is worth
    # +3.1 %: graph.WrwSide, profiles/r06_wrw_side_ab.txt).
    if self.serialise:                                     # measurement mode: nothing of this batch runs beside step k
      st.side_waits_for_main()
    with st.on_side(), profiling.suspended():
      # the iterator itself may enqueue device work (pinned upload + resize kernel of the TFRecord reader run on the CURRENT stream):
      # it has to be the side stream, or the teacher would read a batch the main stream has not finished writing
      images, labels = fetch_raw(lrn)
      x, y = lrn.to_device(images, labels)
      logits = teacher.calc_logits(None, x)
      ev = st.record()
    for t in (images, x, y, logits):
      st.hand_to_main(t)
    self.pending = (x, y, logits, ev, images)
    self.n_issued += 1

  def take(self):
    x, y, logits, ev, __ = self.pending
    self.pending = None
    if ev is not None:
      self.streams.main_waits(ev)
    self.n_taken += 1
    return x, y, logits

  def take_images(self):
    """The prefetched batch as the iterator delivered it (for a consumer of the training iterator other than train_step: it must see
    the batch train_step would have seen, or the data order differs from the run without this helper)."""
    images, ev = self.pending[4], self.pending[3]
    self.pending = None
    if ev is not None:                               # (None: handed over by a suspended step graph, already in main-stream order)
      self.streams.main_waits(ev)
    self.n_taken += 1
    return images

  def drop(self):
    """Forget the prefetched batch (the data iterator was reset: the next step starts from the iterator again)."""
    if self.pending is not None:
      if self.pending[3] is not None:
        self.streams.main_waits(self.pending[3])     # nothing of it may still be running when its memory is released
      self.pending = None


def teacher_of(learner):
  """The learner's DistillationHelper (`helper_dst`; the channel-pruning learner calls it `learner_dst`, as the reference does), or None."""
  return getattr(learner, 'helper_dst', None) or getattr(learner, 'learner_dst', None)


def make(learner):
  """A TeacherAhead for `learner`, or None: distillation only, and a HIP device (or PF_TEACHER_AHEAD=inline: the in-order stand-in
  for the CPU tests)."""
  if not enabled() or teacher_of(learner) is None:
    return None
  dev = learner.device
  if os.environ.get('PF_TEACHER_AHEAD') == 'inline':
    return TeacherAhead(learner, InlineStreams())
  if torch.device(dev).type != 'cuda':
    return None
  return TeacherAhead(learner, CudaStreams(dev))


def of(learner):
  """The learner's helper (made on first use), or None."""
  if not hasattr(learner, '_teacher_ahead'):
    learner._teacher_ahead = make(learner)
  return learner._teacher_ahead


def next_batch(learner):
  """(helper, x, y, teacher logits or None) for the step that starts now: what the previous step issued ahead, or -- first step,
  helper off -- the next batch of the iterator with the teacher still to run in line.  A step that is being recorded into / replayed
  from a hipGraph (step_graph.py) gets the graph's static buffers instead."""
  static = getattr(learner, '_static_batch', None)
  if static is not None:
    return (None,) + tuple(static)
  ahead = of(learner)
  if ahead is not None and ahead.pending is not None:
    taken = ahead.take()
    if ahead.handover_only:                          # PF_TEACHER_AHEAD=0: the hand-over is done, the helper does not stay (ADVICE r4:
      learner._teacher_ahead = ahead = None          # it used to, and every later eager step ran the teacher one step ahead)
    return (ahead,) + taken
  images, labels = fetch_raw(learner)
  x, y = learner.to_device(images, labels)
  return ahead, x, y, None


def next_images(learner):
  """iter_train.get_next()[0] for every consumer of the training iterator that is not train_step (layer-wise tuning): the batch a
  previous step prefetched comes first."""
  sg = getattr(learner, '_step_graph', None)
  if sg is not None:
    sg.yield_to_eager()                              # a ready step graph holds the next batches: they come first
  ahead = of(learner)
  if ahead is not None and ahead.pending is not None and ahead.pending[4] is not None:
    images = ahead.take_images()
    if ahead.handover_only:
      learner._teacher_ahead = None
    return images
  return fetch_raw(learner)[0]


def drop(learner):
  """Forget the prefetched batch, if any, and the batches a suspended step graph handed back: call where the training iterator
  is reset or re-built (they belong to its previous pass)."""
  ahead = getattr(learner, '_teacher_ahead', None)
  if ahead is not None:
    ahead.drop()
    if ahead.handover_only:
      learner._teacher_ahead = None
  back = getattr(learner, '_unget', None)
  if back:
    del back[:]
  sg = getattr(learner, '_step_graph', None)
  if sg is not None:
    sg.discard_lookahead()                           # a ready step graph holds two more batches of that pass in its static buffers
