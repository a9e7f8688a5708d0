"""Export a physically shrunk model from a channel-pruned checkpoint (the step AFTER the path; reference
tools/conversion/export_chn_pruned_tflite_model.py:184-276).

The learners prune "fake": tensors keep their shape and pruned input channels are all-zero slices of the kernel.  As
in the reference, every Conv2D whose kernel has all-zero input channels is replaced by

    gather(input, nnz, axis=channel)  ->  convolution with kernel[:, :, nnz, :]

(`graph_trans_mthd='gather'`; the reference's TF-Lite variant expresses the gather as a 1x1 one-hot convolution).
The artefact is a plain `.npz`: every variable of the checkpoint, with `<conv>/kernel` shrunk to [kh, kw, nnz, cout]
(HWIO, like the reference's checkpoints) and an extra int32 `<conv>/kernel/gather` index vector for each shrunk
convolution, plus `export_summary.json` (per-layer channels and multiply-accumulates before / after).

    python -m pocketflow_amd.tools.conversion.export_chn_pruned_model --model_dir ./models_cpg_eval \\
        [--enbl_fake_prune --fake_prune_ratio 0.5]      # random pruning, for speed tests only (reference :184-201)

No device is needed: each replacement is verified against the original convolution on random inputs with torch-CPU
(`verify_layer`), which is also the executable definition of how to run the artefact.
"""
from __future__ import annotations

import json
import logging
import os
import sys
from typing import Dict, Tuple

import numpy as np
import torch

from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.utils import checkpoint

flags.DEFINE_string('model_dir', './models', 'model directory')
flags.DEFINE_boolean('enbl_fake_prune', False, 'enable fake pruning (for speed test only)')
flags.DEFINE_float('fake_prune_ratio', 0.5, 'fake pruning ratio')
flags.DEFINE_string('export_file', 'model_shrunk.npz', 'file name of the exported model (inside model_dir)')

log = logging.getLogger('pocketflow_amd')


def is_conv_kernel(name: str, value: np.ndarray) -> bool:
  """Kernels read by a Conv2D op: 4-D, not a depthwise kernel (`depthwise_weights`, [kh, kw, C, 1] applied per channel)."""
  return value.ndim == 4 and 'depthwise' not in name and (name.endswith('/kernel') or name.endswith('/weights'))


def apply_fake_pruning(kernel: np.ndarray, rng: np.random.RandomState) -> np.ndarray:
  """Zero a random `fake_prune_ratio` of the input channels (reference :184-201)."""
  nb_chns = kernel.shape[2]
  idxs_all = np.arange(nb_chns)
  rng.shuffle(idxs_all)
  kernel = kernel.copy()
  kernel[:, :, idxs_all[:int(nb_chns * FLAGS.fake_prune_ratio)], :] = 0.0
  return kernel


def shrink_kernel(kernel: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
  """(kernel[:, :, nnz, :], nnz) with nnz the input channels whose slice is not all-zero (reference :236-241)."""
  nnzs = np.nonzero(np.sum(np.abs(kernel), axis=(0, 1, 3)))[0].astype(np.int32)
  return np.ascontiguousarray(kernel[:, :, nnzs, :]), nnzs


def conv_gather(x_nchw: torch.Tensor, kernel_hwio: np.ndarray, gather: np.ndarray = None, stride: int = 1, padding=0):
  """How an exported convolution runs: gather the surviving input channels, then the smaller convolution."""
  w = torch.from_numpy(np.ascontiguousarray(np.transpose(kernel_hwio, (3, 2, 0, 1)))).to(x_nchw.dtype)
  if gather is not None:
    x_nchw = x_nchw.index_select(1, torch.from_numpy(np.asarray(gather, dtype=np.int64)))
  return torch.nn.functional.conv2d(x_nchw, w, None, stride=stride, padding=padding)


def verify_layer(kernel: np.ndarray, shrunk: np.ndarray, gather: np.ndarray, rng: np.random.RandomState) -> float:
  """max |conv(x, kernel) - conv(gather(x), shrunk)| on a random input (exactly 0 up to summation order)."""
  x = torch.from_numpy(rng.randn(2, kernel.shape[2], 9, 9).astype(np.float32))
  pad = kernel.shape[0] // 2
  ref = conv_gather(x, kernel, None, 1, pad)
  got = conv_gather(x, shrunk, gather, 1, pad)
  return float((ref - got).abs().max())


def export(values: Dict[str, np.ndarray], rng=None):
  """values: checkpoint variables in the reference layout -> (exported variables, summary rows)."""
  rng = rng or np.random.RandomState(0)
  out, rows = {}, []
  for name, val in values.items():
    if not is_conv_kernel(name, val):
      out[name] = val
      continue
    kernel = apply_fake_pruning(val, rng) if FLAGS.enbl_fake_prune else val
    shrunk, nnzs = shrink_kernel(kernel)
    cin = kernel.shape[2]
    if nnzs.size == cin:
      out[name] = kernel
    elif nnzs.size == 0:
      raise ValueError('%s: every input channel is pruned' % name)
    else:
      err = verify_layer(kernel, shrunk, nnzs, rng)
      if err > 1e-4 * max(1.0, float(np.abs(kernel).max()) * cin):
        raise AssertionError('%s: shrunk convolution differs from the original by %g' % (name, err))
      out[name] = shrunk
      out[name + '/gather'] = nnzs
      log.info('reducing %d channels to %d: %s' % (cin, nnzs.size, name))
    rows.append({'name': name, 'kh': int(kernel.shape[0]), 'kw': int(kernel.shape[1]), 'cin': int(cin),
                 'cin_kept': int(nnzs.size), 'cout': int(kernel.shape[3]),
                 'macs_per_pixel': int(np.prod(kernel.shape)), 'macs_per_pixel_kept': int(np.prod(out[name].shape))})
  return out, rows


def load_exported(path: str) -> Dict[str, np.ndarray]:
  with np.load(path) as f:
    return {k.replace('|', '/'): f[k] for k in f.files}


def main(argv=None):
  FLAGS.parse(argv if argv is not None else sys.argv[1:])
  logging.basicConfig(level=logging.INFO)
  prefix = checkpoint.latest_checkpoint(FLAGS.model_dir)
  if prefix is None:
    raise FileNotFoundError('no checkpoint under ' + FLAGS.model_dir)
  values = checkpoint.load(prefix)
  out, rows = export(values)
  path = os.path.join(FLAGS.model_dir, FLAGS.export_file)
  np.savez(path, **{k.replace('/', '|'): v for k, v in out.items()})
  before, after = sum(r['macs_per_pixel'] for r in rows), sum(r['macs_per_pixel_kept'] for r in rows)
  with open(os.path.join(FLAGS.model_dir, 'export_summary.json'), 'w') as f:
    json.dump({'source': prefix, 'layers': rows, 'kernel_params': before, 'kernel_params_kept': after}, f, indent=1)
  log.info('%s generated: %d of %d convolution-kernel parameters kept (%.1f %%)' % (path, after, before, 100.0 * after / max(before, 1)))
  return 0


if __name__ == '__main__':
  sys.exit(main())
