"""Export the INTEGER form of a uniformly quantised model (the step AFTER the path).

The reference ships quantised models through TF-Lite (tools/conversion/export_quant_tflite_model.py:196-249: freeze the
graph, hand it to the TF-Lite converter) -- for its `uniform-tf` learner only, whose fake-quant nodes the converter
understands; the UniformQuantLearner of the hot path (learners/uniform_quantization) keeps float32 master weights in its
checkpoints and has no integer artefact at all.  This tool gives it one, defined by the learner's OWN quantiser
(`__uniform_quantize`, learners/uniform_quantization/utils.py:163-245): for every quantised matmul kernel

    alpha = max - min + 1e-10,  beta = min        per tensor / per output channel / per (strided) split bucket
    code  = round((w - beta) / alpha * k)         k = 2^bits - 1, an integer in [0, k]
    w_q   = alpha * (code / k) + beta             float32, mul then add: the reference's __inv_scale, bit for bit

so `decode_tensor` reproduces the fake-quantised weights the training graph multiplied with EXACTLY, and a model restored
from the artefact evaluates identically (re-quantising a quantised tensor is the identity).  Where the numbers come from:
alpha / beta and the fake-quantised weights are taken from the device kernels the learner trains with (`pf_seg_minmax`,
`pf_seg_uq_apply`, float32), the codes are read back from them and every tensor is verified to round-trip bit-exactly
before it is written -- there is no second implementation of the quantiser here.

Artefact: one `.npz` -- per quantised kernel `<name>/codes` (bit-packed, `bits` per weight, HWIO order like the reference's
checkpoints), `<name>/alpha`, `<name>/beta` (float32, one per bucket), `<name>/meta` = [bits, bucket mode, bucket size, *shape];
every other variable as float32 -- plus `export_summary.json` (bytes before / after per tensor, incl. the 2 x 32 bits per
bucket the reference accounts for, uq utils.py:299-306).

    python -m pocketflow_amd.tools.conversion.export_quant_int8_model --model_name resnet --dataset_name ilsvrc_12 \\
        --uql_weight_bits 8 [--uql_use_buckets --uql_bucket_type channel]      # reads uql_save_quant_model_path (needs the GPU)
"""
from __future__ import annotations

import json
import logging
import os
from typing import Dict, Tuple

import numpy as np

log = logging.getLogger('pocketflow_amd')

MODE_TENSOR, MODE_CHANNEL, MODE_SPLIT = 0, 1, 2


def _bucket_index(shape, mode: int, bucket_size: int) -> np.ndarray:
  """Bucket of every element of an HWIO-ordered tensor, flattened: whole tensor | output channel (the last axis of
  reshape(w, [-1, cout]), uq utils.py:277-289) | the COLUMN of reshape(padded flat, [bucket_size, multiple]) with multiple =
  ceil(n / bucket_size) buckets (:247-275) -- a 'split' bucket is the strided set {flat[i * multiple + j]}_i."""
  n = int(np.prod(shape))
  if mode == MODE_TENSOR:
    return np.zeros(n, np.int64)
  if mode == MODE_CHANNEL:
    return np.tile(np.arange(shape[-1], dtype=np.int64), n // shape[-1])
  return np.arange(n, dtype=np.int64) % (-(-n // bucket_size))


def pack_codes(codes: np.ndarray, bits: int) -> np.ndarray:
  """`bits` per code, little-endian bit order, padded to a whole byte."""
  b = ((codes.reshape(-1, 1).astype(np.uint8) >> np.arange(bits, dtype=np.uint8)) & 1).astype(np.uint8)
  return np.packbits(b.reshape(-1), bitorder='little')


def unpack_codes(packed: np.ndarray, bits: int, n: int) -> np.ndarray:
  b = np.unpackbits(packed, bitorder='little')[:n * bits].reshape(n, bits).astype(np.uint16)
  return (b << np.arange(bits, dtype=np.uint16)).sum(axis=1).astype(np.uint8)


def decode_tensor(packed: np.ndarray, alpha: np.ndarray, beta: np.ndarray, meta: np.ndarray) -> np.ndarray:
  """The executable definition of the format: float32 HWIO tensor from codes + per-bucket (alpha, beta)."""
  bits, mode, bucket_size = int(meta[0]), int(meta[1]), int(meta[2])
  shape = tuple(int(v) for v in meta[3:])
  n = int(np.prod(shape))
  codes = unpack_codes(packed, bits, n).astype(np.float32)
  k = np.float32(np.int64(2) ** np.int64(bits) - np.int64(1))
  idx = _bucket_index(shape, mode, bucket_size)
  q = (codes / k).astype(np.float32)
  return ((alpha.astype(np.float32)[idx] * q).astype(np.float32) + beta.astype(np.float32)[idx]).astype(np.float32).reshape(shape)


def encode_tensor(wq: np.ndarray, alpha: np.ndarray, beta: np.ndarray, bits: int, mode: int,
                  bucket_size: int) -> Tuple[np.ndarray, np.ndarray]:
  """Codes of an already fake-quantised HWIO tensor `wq` under the quantiser's own (alpha, beta); raises unless
  decode(encode(wq)) == wq bit for bit."""
  if not 1 <= bits <= 8:
    raise ValueError('integer export covers 1..8 bits (got %d)' % bits)
  shape = wq.shape
  idx = _bucket_index(shape, mode, bucket_size)
  k = float(2 ** bits - 1)
  a, b = alpha.astype(np.float64)[idx], beta.astype(np.float64)[idx]
  codes = np.clip(np.rint((wq.reshape(-1).astype(np.float64) - b) / a * k), 0, k).astype(np.uint8)
  meta = np.array([bits, mode, bucket_size] + list(shape), np.int64)
  packed = pack_codes(codes, bits)
  back = decode_tensor(packed, alpha, beta, meta)
  if not np.array_equal(back.view(np.uint32), np.ascontiguousarray(wq, np.float32).view(np.uint32)):
    bad = int((back != wq).sum())
    raise ValueError('%d of %d weights do not round-trip through their integer codes' % (bad, wq.size))
  return packed, meta


def export_from_learner(learner, path: str) -> Dict:
  """Artefact of a UniformQuantLearner in its current state (rank 0).  Returns the summary dict."""
  import torch
  g = learner.graph
  st, plan, uq = g.store, learner.uni_quant.plan, learner.uni_quant
  qw = torch.empty_like(st.w_master)                      # float32 fake-quantised copy, whatever the compute dtype is
  plan.uniform_quantize(st.w_master, qw)
  ab = plan.alpha_beta().cpu().numpy()
  qw = qw.cpu().numpy()
  seg_bits = plan.segs_host['bits']
  out, rows = {}, []
  quantised = set()
  for s, (v, w) in enumerate(zip(uq._all_vars, plan.weights)):
    bits = int(seg_bits[s])
    if bits <= 0 or bits > 8:
      continue
    nb, so = plan.n_buckets[s], plan.slot_offsets[s]
    if not plan.use_buckets or nb == 1 and plan.bucket_type == 'channel':
      mode = MODE_TENSOR
    else:
      mode = MODE_CHANNEL if plan.bucket_type == 'channel' else MODE_SPLIT
    wq = v.to_ref(qw[v.offset:v.offset + v.numel])
    packed, meta = encode_tensor(wq, ab[so:so + nb, 0], ab[so:so + nb, 1], bits, mode, int(plan.bucket_size))
    out[v.name + '/codes'], out[v.name + '/alpha'], out[v.name + '/beta'] = packed, ab[so:so + nb, 0].copy(), ab[so:so + nb, 1].copy()
    out[v.name + '/meta'] = meta
    quantised.add(v.name)
    rows.append(dict(name=v.name, bits=bits, buckets=int(nb), float32_bytes=int(wq.size * 4),
                     int_bytes=int(packed.nbytes + nb * 8)))
  for name, val in st.export_numpy().items():
    if name not in quantised:
      out[name] = np.asarray(val, np.float32)
  os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
  np.savez(path, **out)
  f32 = sum(r['float32_bytes'] for r in rows)
  summ = dict(quantised_tensors=len(rows), float32_bytes=f32, int_bytes=sum(r['int_bytes'] for r in rows),
              other_float32_bytes=int(sum(v.nbytes for k, v in out.items() if k.split('/')[-1] not in ('codes', 'alpha', 'beta', 'meta'))),
              layers=rows)
  with open(os.path.join(os.path.dirname(os.path.abspath(path)), 'export_summary.json'), 'w') as f:
    json.dump(summ, f, indent=1)
  log.info('integer model written to %s: %d tensors, %.2f MB -> %.2f MB', path, len(rows), f32 / 1e6, summ['int_bytes'] / 1e6)
  return summ


def load_exported(path: str) -> Dict[str, np.ndarray]:
  """{variable name: float32 array} with every quantised kernel decoded: what `restore_vars` / `load_numpy` take."""
  z = np.load(path)
  out, done = {}, set()
  for key in z.files:
    if key.endswith('/codes'):
      name = key[:-len('/codes')]
      out[name] = decode_tensor(z[key], z[name + '/alpha'], z[name + '/beta'], z[name + '/meta'])
      done.update({key, name + '/alpha', name + '/beta', name + '/meta'})
  for key in z.files:
    if key not in done:
      out[key] = z[key]
  return out


def main(argv=None):
  import sys
  from pocketflow_amd.flags import FLAGS, flags
  flags.DEFINE_string('export_file', 'model_int8.npz', 'file name of the exported model (beside uql_save_quant_model_path)')
  flags.DEFINE_string('model_name', 'resnet', 'lenet | resnet | mobilenet')
  flags.DEFINE_string('dataset_name', 'cifar_10', 'cifar_10 | ilsvrc_12')
  import importlib
  import pocketflow_amd.learners.uniform_quantization.learner as UQ
  FLAGS.parse(argv if argv is not None else sys.argv[1:])
  mod = importlib.import_module('pocketflow_amd.nets.%s_at_%s' % (FLAGS.model_name, FLAGS.dataset_name.replace('_', '')))
  learner = UQ.UniformQuantLearner(None, mod.ModelHelper())
  from pocketflow_amd.utils import checkpoint
  learner.restore_vars(checkpoint.latest_checkpoint(os.path.dirname(FLAGS.uql_save_quant_model_path)))
  path = os.path.join(os.path.dirname(FLAGS.uql_save_quant_model_path), FLAGS.export_file)
  print(json.dumps({k: v for k, v in export_from_learner(learner, path).items() if k != 'layers'}))


if __name__ == '__main__':
  main()
