"""path.conf -> command-line flags (same contract as the reference's utils/get_path_args.py:22-74).

Usage: python -m pocketflow_amd.utils.get_path_args <local|docker|seven> <net_run.py> <path.conf>
prints e.g. `--model_http_url https://... --data_dir_local /data/cifar-10`.  The dataset key is taken
from the entry script's file name (`..._at_<dataset>_run.py`); `None` values are skipped; `#` starts
a comment.
"""
from __future__ import annotations

import re
import sys


def read_conf(conf_file):
  """Yield (key, value) pairs of a path.conf file."""
  with open(conf_file, 'r') as f:
    for raw in f:
      line = raw.split('#', 1)[0].strip()
      if not line:
        continue
      parts = line.split(' = ')
      assert len(parts) == 2, 'each line must contains exactly one \' = \''
      yield parts[0].strip(), parts[1].strip()


def get_path_args(exec_mode, py_file, conf_file):
  m = re.search(r'at_[0-9A-Za-z]+_run.py$', py_file)
  assert m is not None, 'unable to match pattern in ' + py_file
  dataset = m.group(0).split('_')[1]
  args = []
  data_dirs = {'local': None, 'docker': None, 'seven': None, 'hdfs': None}
  own = re.compile(r'^data_dir_[a-z]+_%s$' % dataset)
  for key, value in read_conf(conf_file):
    if value == 'None':
      continue
    if not key.startswith('data_dir_'):
      args.append('--%s %s' % (key, value))
    elif own.match(key):
      data_dirs[key.split('_')[2]] = value
  if exec_mode in ('local', 'seven') and data_dirs[exec_mode] is not None:
    args.append('--data_dir_local %s' % data_dirs[exec_mode])
  elif data_dirs['local'] is not None:
    args.append('--data_dir_local %s' % data_dirs['docker'])
  if data_dirs['hdfs'] is not None:
    args.append('--data_dir_hdfs %s' % data_dirs['hdfs'])
  return ' '.join(args)


if __name__ == '__main__':
  assert len(sys.argv) == 4
  print(get_path_args(sys.argv[1], sys.argv[2], sys.argv[3]))
