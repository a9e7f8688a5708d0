"""Checkpoint files: tf.train.Saver stand-in.

`save(values, save_path, global_step)` writes `<save_path>-<step>.npz` holding every variable under
its TF-style name in the REFERENCE layout (HWIO kernels, [in,out] dense), plus a `checkpoint` index
file in the same directory; `latest_checkpoint(dir)` returns the newest prefix, like
tf.train.latest_checkpoint.  (Reading real TF-Saver-V2 archives is a SURVEY 8f "next" row.)
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np


def save(values: Dict[str, np.ndarray], save_path: str, global_step: Optional[int] = None) -> str:
  d = os.path.dirname(save_path)
  if d:
    os.makedirs(d, exist_ok=True)
  prefix = save_path if global_step is None else '%s-%d' % (save_path, int(global_step))
  np.savez(prefix + '.npz', **{k.replace('/', '|'): v for k, v in values.items()})
  with open(os.path.join(d, 'checkpoint'), 'w') as f:
    f.write('model_checkpoint_path: "%s"\n' % os.path.basename(prefix))
  return prefix


def latest_checkpoint(ckpt_dir: str) -> Optional[str]:
  index = os.path.join(ckpt_dir, 'checkpoint')
  if not os.path.exists(index):
    return None
  with open(index, 'r') as f:
    line = f.readline().strip()
  name = line.split('"')[1]
  prefix = os.path.join(ckpt_dir, name)
  return prefix if os.path.exists(prefix + '.npz') else None


def load(prefix: str) -> Dict[str, np.ndarray]:
  with np.load(prefix + '.npz') as z:
    return {k.replace('|', '/'): z[k] for k in z.files}
