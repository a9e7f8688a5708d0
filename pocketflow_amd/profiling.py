"""Optional HIP-event timing of named kernel groups (used by bench.py for the roofline figure).

`with region('bn_bwd_apply'):` records a start/stop event pair on the CURRENT stream (the stream the
kernels are launched on) when that name is enabled; otherwise it costs one dict lookup.
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import Dict, List, Tuple

import torch

_enabled: Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event, float]]] = {}


def enable(name: str) -> None:
  _enabled[name] = []


def disable_all() -> None:
  _enabled.clear()


def reset() -> None:
  for k in _enabled:
    _enabled[k] = []


@contextmanager
def region(name: str, work: float = 0.0):
  rec = _enabled.get(name)
  if rec is None:
    yield
    return
  a = torch.cuda.Event(enable_timing=True)
  b = torch.cuda.Event(enable_timing=True)
  a.record()
  yield
  b.record()
  rec.append((a, b, work))


def summary(name: str):
  """(launches, total_ms, total_work) of a region; call after torch.cuda.synchronize()."""
  rec = _enabled.get(name, [])
  ms = sum(a.elapsed_time(b) for a, b, _ in rec)
  work = sum(w for _, _, w in rec)
  return len(rec), ms, work
