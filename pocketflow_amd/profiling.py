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
  _enabled.clear()


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
  __slots__ = ('rec', 'work', 'a')

  def __init__(self, rec, work):
    self.rec, self.work, self.a = rec, work, None

  def __enter__(self):
    self.a = torch.cuda.Event(enable_timing=True)
    self.a.record()
    return None

  def __exit__(self, exc_type, exc, tb):
    if exc_type is None:
      b = torch.cuda.Event(enable_timing=True)
      b.record()
      self.rec.append((self.a, b, self.work))
    return False


_NOOP = _Noop()
_suspended = 0
_paused = False


class suspended(object):
  """`with suspended():` -- regions entered inside record nothing (launches issued on a side stream beside other work: an
  event-to-event interval there is not a kernel's duration, learners/teacher_ahead.py)."""

  def __enter__(self):
    global _suspended
    _suspended += 1
    return None

  def __exit__(self, exc_type, exc, tb):
    global _suspended
    _suspended -= 1
    return False


def region(name: str, work: float = 0.0):
  rec = _enabled.get(name)
  return _NOOP if (rec is None or _suspended or _paused) else _Timed(rec, work)


def count(name: str) -> int:
  """Launches recorded so far (host side; no synchronisation)."""
  return len(_enabled.get(name, []))


def summary(name: str, lo: int = 0, hi: int = None):
  """(launches, total_ms, total_work) of a region (records lo .. hi); call after torch.cuda.synchronize()."""
  rec = _enabled.get(name, [])[lo:hi]
  ms = sum(a.elapsed_time(b) for a, b, _ in rec)
  work = sum(w for _, _, w in rec)
  return len(rec), ms, work
