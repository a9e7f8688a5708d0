"""Checkpoint files: tf.train.Saver stand-in.

`save(values, save_path, global_step)` writes every variable under its TF-style name in the REFERENCE layout
(HWIO kernels, [in,out] dense) -- as `<prefix>.npz` (default) or as a TensorFlow Saver-V2 bundle
(`fmt='tf'`: `<prefix>.index` + `<prefix>.data-00000-of-00001`, see tf_checkpoint.py) -- plus the `checkpoint`
state file tf.train.latest_checkpoint reads; `load(prefix)` reads either, so the pre-trained archives of the
reference (`models_<model>_at_<dataset>.tar.gz` -> `./models/model.ckpt-*.index`) restore directly.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np


def save(values: Dict[str, np.ndarray], save_path: str, global_step: Optional[int] = None, fmt: str = 'npz') -> str:
  d = os.path.dirname(save_path)
  if d:
    os.makedirs(d, exist_ok=True)
  prefix = save_path if global_step is None else '%s-%d' % (save_path, int(global_step))
  if fmt == 'tf':
    from pocketflow_amd.utils import tf_checkpoint
    tf_checkpoint.write_bundle(values, prefix)
  elif fmt == 'npz':
    np.savez(prefix + '.npz', **{k.replace('/', '|'): v for k, v in values.items()})
  else:
    raise ValueError("checkpoint format must be 'npz' or 'tf', not %r" % fmt)
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
  return prefix if (os.path.exists(prefix + '.npz') or os.path.exists(prefix + '.index')) else None


def load(prefix: str) -> Dict[str, np.ndarray]:
  if prefix is None:
    raise FileNotFoundError('no checkpoint to restore: latest_checkpoint() found none (missing or stale `checkpoint` '
                            'state file) -- was download_model() / a save run for this directory?')
  if not os.path.exists(prefix + '.npz') and os.path.exists(prefix + '.index'):
    from pocketflow_amd.utils import tf_checkpoint
    return tf_checkpoint.read_bundle(prefix)
  with np.load(prefix + '.npz') as z:
    return {k.replace('|', '/'): z[k] for k in z.files}
