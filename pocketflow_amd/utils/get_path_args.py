"""`path.conf` -> command-line flags (reference utils/get_path_args.py:22-74; same CLI:
`python -m pocketflow_amd.utils.get_path_args <local|docker|seven> <net_run.py> <path.conf>`).

`key = value` lines, `#` comments, `None` = unset.  The dataset name comes from the run script's
file name (`..._at_<dataset>_run.py`); `data_dir_<disk>_<dataset>` keys select the data directory
for the execution mode, every other key is passed through as `--key value`."""
from __future__ import annotations

import re
import sys
from typing import List


def get_path_args(exec_mode: str, py_file: str, conf_file: str) -> str:
  found = re.search(r'at_[0-9A-Za-z]+_run.py$', py_file)
  assert found is not None, 'unable to match pattern in ' + py_file
  dataset_name = found.group(0).split('_')[1]
  key_re = re.compile(r'^data_dir_[a-z]+_%s$' % dataset_name)
  data_dirs = {'local': None, 'docker': None, 'seven': None, 'hdfs': None}
  args: List[str] = []
  with open(conf_file, 'r') as f:
    for line in f:
      line = line.split('#', 1)[0].strip()
      if not line:
        continue
      parts = line.split(' = ')
      assert len(parts) == 2, "each line must contains exactly one ' = '"
      key, value = parts[0].strip(), parts[1].strip()
      if value == 'None':
        continue
      if not key.startswith('data_dir_'):
        args.append('--%s %s' % (key, value))
      elif key_re.match(key):
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
