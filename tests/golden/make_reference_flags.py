#!/usr/bin/env python
"""tests/golden/reference_flags.json: every tf.app.flags definition of the reference files on (or next to) the hot path
-- name, kind, default, defining file(s).  Build container only (reads /root/reference); no code is executed, the
definitions are read with a regular expression and the defaults evaluated as Python literals."""
import ast
import glob
import json
import os
import re

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
SKIP = ('channel_pruning_rmt', 'discr_channel_pruning', 'uniform_quantization_tf', 'pascalvoc', 'vgg_at', 'faster', 'ssd')


def main():
  files = []
  for pat in ('learners/*.py', 'learners/*/*.py', 'nets/*.py', 'datasets/*.py', 'utils/*.py', 'rl_agents/ddpg/*.py', 'main.py'):
    files += glob.glob(os.path.join(REF, pat))
  flags = {}
  for f in sorted(files):
    if any(s in f for s in SKIP):
      continue
    for m in re.finditer(r"DEFINE_(\w+)\(\s*'(\w+)'\s*,\s*(?:\\\s*)?([^,\n]+)", open(f).read()):
      kind, name, default = m.group(1), m.group(2), m.group(3).strip()
      try:
        value = ast.literal_eval(default)
      except (ValueError, SyntaxError):
        value = default
      flags.setdefault(name, []).append({'kind': kind, 'default': value, 'file': os.path.relpath(f, REF)})
  with open(os.path.join(HERE, 'reference_flags.json'), 'w') as o:
    json.dump(flags, o, indent=1, sort_keys=True)
  print('%d flags' % len(flags))


if __name__ == '__main__':
  main()
