"""`tf.app.flags` stand-in (absl is absent): same DEFINE_* calls, same global FLAGS object.

The reference defines flags at import time in whichever module needs them, reads
them as `FLAGS.name`, and treats unknown command-line flags as fatal.  This module reproduces
that contract: `from pocketflow_amd.flags import flags, FLAGS` then `flags.DEFINE_integer(...)`.
"""
from __future__ import annotations

import sys
from typing import Any, Dict, List, Optional


class _FlagValues:
  def __init__(self):
    object.__setattr__(self, '_defs', {})
    object.__setattr__(self, '_vals', {})
    object.__setattr__(self, '_explicit', set())

  def _define(self, name: str, default: Any, help_: str, kind: str) -> None:
    # The reference defines e.g. `batch_size` in whichever dataset module the entry script imports
    # (cifar10: 128, ilsvrc12: 64).  In one process that imports several, the latest definition
    # provides the default unless the user set the flag explicitly.
    self._defs[name] = (kind, default, help_)
    if name not in self._explicit:
      self._vals[name] = default

  def __getattr__(self, name: str) -> Any:
    vals = object.__getattribute__(self, '_vals')
    if name in vals:
      return vals[name]
    raise AttributeError('Unknown command line flag %r' % name)

  def __setattr__(self, name: str, value: Any) -> None:
    if name not in self._defs:
      raise AttributeError('Unknown command line flag %r' % name)
    self._vals[name] = self._convert(name, value)
    self._explicit.add(name)

  def __contains__(self, name: str) -> bool:
    return name in self._defs

  def _convert(self, name: str, value: Any) -> Any:
    kind = self._defs[name][0]
    if value is None or not isinstance(value, str):
      return value
    if kind == 'integer':
      return int(value)
    if kind == 'float':
      return float(value)
    if kind == 'boolean':
      if value.lower() in ('true', 't', '1', 'yes'):
        return True
      if value.lower() in ('false', 'f', '0', 'no'):
        return False
      raise ValueError('flag --%s: bad boolean %r' % (name, value))
    return None if value == 'None' else value

  def parse(self, argv: Optional[List[str]] = None) -> List[str]:
    """Parse `--name value`, `--name=value`, `--flag` / `--noflag`; unknown flags are fatal."""
    argv = list(sys.argv[1:] if argv is None else argv)
    rest, i = [], 0
    while i < len(argv):
      a = argv[i]
      if not a.startswith('--'):
        rest.append(a)
        i += 1
        continue
      body = a[2:]
      if '=' in body:
        name, val = body.split('=', 1)
      else:
        name, val = body, None
      if name not in self._defs and name.startswith('no') and name[2:] in self._defs \
          and self._defs[name[2:]][0] == 'boolean':
        self._vals[name[2:]] = False
        self._explicit.add(name[2:])
        i += 1
        continue
      if name not in self._defs:
        raise ValueError('Unknown command line flag %r' % name)
      if val is None:
        if self._defs[name][0] == 'boolean' and (i + 1 >= len(argv) or argv[i + 1].startswith('--')):
          val = 'true'
        else:
          i += 1
          if i >= len(argv):
            raise ValueError('flag --%s needs a value' % name)
          val = argv[i]
      self._vals[name] = self._convert(name, val)
      self._explicit.add(name)
      i += 1
    return rest

  def reset(self) -> None:
    for name, (_, default, _h) in self._defs.items():
      self._vals[name] = default
    self._explicit.clear()

  def flag_values_dict(self) -> Dict[str, Any]:
    return dict(self._vals)


FLAGS = _FlagValues()


class _Flags:
  FLAGS = FLAGS

  @staticmethod
  def DEFINE_string(name, default, help_=''):
    FLAGS._define(name, default, help_, 'string')

  @staticmethod
  def DEFINE_integer(name, default, help_=''):
    FLAGS._define(name, default, help_, 'integer')

  @staticmethod
  def DEFINE_float(name, default, help_=''):
    FLAGS._define(name, default, help_, 'float')

  @staticmethod
  def DEFINE_boolean(name, default, help_=''):
    FLAGS._define(name, default, help_, 'boolean')

  DEFINE_bool = DEFINE_boolean


flags = _Flags()
