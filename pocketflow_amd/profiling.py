"""Optional HIP-event timing of named kernel groups (used by bench.py for the roofline figure).

`with region('bn_bwd_apply'):` records a start/stop event pair on the CURRENT stream (the stream the
kernels are launched on) when that name is enabled; otherwise it costs one dict lookup.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

_enabled: Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event, float]]] = {}


def enable(name: str) -> None:
  global _paused
  _paused = False
  _enabled[name] = []


def pause() -> None:
  """Stop recording (what was recorded stays for `summary`)."""
  global _paused
  _paused = True


def unpause() -> None:
  global _paused
  _paused = False


def disable_all() -> None:
  global include_side, _paused
  _enabled.clear()
  include_side = False
  _paused = False


def reset() -> None:
  for k in _enabled:
    _enabled[k] = []


class _Noop(object):
  """What `region` returns for a name that is not enabled: entering and leaving costs two trivial calls (the generator-based
  context manager this replaces cost ~1.2 us per launch site, a few hundred sites per step)."""
  __slots__ = ()

  def __enter__(self):
    return None

  def __exit__(self, exc_type, exc, tb):
    return False


class _Timed(object):
  __slots__ = ('rec', 'work', 'a', 'side')

  def __init__(self, rec, work):
    self.rec, self.work, self.a, self.side = rec, work, None, _in_side > 0

  def __enter__(self):
    self.a = torch.cuda.Event(enable_timing=True)
    self.a.record()
    return None

  def __exit__(self, exc_type, exc, tb):
    if exc_type is None:
      b = torch.cuda.Event(enable_timing=True)
      b.record()
      self.rec.append((self.a, b, self.work, self.side))
    return False


_NOOP = _Noop()
_suspended = 0
_in_side = 0
include_side = False       # bench.py: regions entered on the teacher's side stream are recorded too (their event-to-event durations
#                            include the time-sharing with the main stream -- exactly what a rocprofv3 kernel trace shows for them)
_paused = False


class suspended(object):
  """`with suspended():` -- regions entered inside record nothing (launches issued on a side stream beside other work: an
  event-to-event interval there is not a kernel's duration, learners/teacher_ahead.py)."""

  def __enter__(self):
    global _suspended, _in_side
    _in_side += 1
    if not include_side:
      _suspended += 1
    return None

  def __exit__(self, exc_type, exc, tb):
    global _suspended, _in_side
    _in_side -= 1
    if not include_side:
      _suspended -= 1
    return False


def region(name: str, work: float = 0.0):
  rec = _enabled.get(name)
  return _NOOP if (rec is None or _suspended or _paused) else _Timed(rec, work)


def count(name: str) -> int:
  """Launches recorded so far (host side; no synchronisation)."""
  return len(_enabled.get(name, []))


def summary(name: str, lo: int = 0, hi: int = None, side=None):
  """(launches, total_ms, total_work) of a region (records lo .. hi; side: None = all, False = main stream only, True = launches
  issued inside `suspended()` with `include_side`, i.e. the teacher's side stream); call after torch.cuda.synchronize()."""
  rec = [r for r in _enabled.get(name, [])[lo:hi] if side is None or r[3] == side]
  ms = sum(r[0].elapsed_time(r[1]) for r in rec)
  work = sum(r[2] for r in rec)
  return len(rec), ms, work
