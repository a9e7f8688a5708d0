"""tools/conversion: physical channel removal from a fake-pruned checkpoint (reference
tools/conversion/export_chn_pruned_tflite_model.py:184-276).  No GPU."""
import json
import os

import numpy as np
import torch


def test_export_channel_pruned_model(tmp_path):
  from pocketflow_amd.tools.conversion import export_chn_pruned_model as E
  from pocketflow_amd.utils import checkpoint
  rng = np.random.RandomState(1)
  k1 = rng.randn(3, 3, 8, 16).astype(np.float32)
  k1[:, :, [1, 4, 5], :] = 0
  k2 = rng.randn(1, 1, 16, 4).astype(np.float32)                     # nothing pruned
  dw = rng.randn(3, 3, 16, 1).astype(np.float32)
  dw[:, :, 3, :] = 0                                                 # depthwise kernels are never shrunk
  values = {'model/conv1/kernel': k1, 'model/Conv2d_1_pointwise/weights': k2, 'model/Conv2d_1_depthwise/depthwise_weights': dw,
            'model/bn/gamma': np.ones(16, np.float32), 'model/dense/kernel': rng.randn(4, 10).astype(np.float32)}
  checkpoint.save(values, str(tmp_path / 'model.ckpt'), 7)
  assert E.main(['--model_dir', str(tmp_path)]) == 0
  out = E.load_exported(str(tmp_path / 'model_shrunk.npz'))
  assert out['model/conv1/kernel'].shape == (3, 3, 5, 16) and out['model/conv1/kernel/gather'].tolist() == [0, 2, 3, 6, 7]
  assert np.array_equal(out['model/conv1/kernel'], k1[:, :, [0, 2, 3, 6, 7], :])
  assert np.array_equal(out['model/Conv2d_1_pointwise/weights'], k2) and 'model/Conv2d_1_pointwise/weights/gather' not in out
  assert np.array_equal(out['model/Conv2d_1_depthwise/depthwise_weights'], dw)
  assert np.array_equal(out['model/dense/kernel'], values['model/dense/kernel'])
  # running the artefact: gather + smaller convolution == the original convolution
  x = torch.from_numpy(rng.randn(2, 8, 6, 6).astype(np.float32))
  ref = E.conv_gather(x, k1, None, 1, 1)
  got = E.conv_gather(x, out['model/conv1/kernel'], out['model/conv1/kernel/gather'], 1, 1)
  assert float((ref - got).abs().max()) < 1e-5
  summ = json.load(open(tmp_path / 'export_summary.json'))
  assert summ['kernel_params'] == k1.size + k2.size and summ['kernel_params_kept'] == 3 * 3 * 5 * 16 + k2.size
  assert os.path.basename(summ['source']) == 'model.ckpt-7'


def test_export_fake_prune_option(tmp_path):
  from pocketflow_amd.tools.conversion import export_chn_pruned_model as E
  from pocketflow_amd.utils import checkpoint
  rng = np.random.RandomState(2)
  checkpoint.save({'model/conv/kernel': rng.randn(3, 3, 10, 6).astype(np.float32)}, str(tmp_path / 'model.ckpt'), 0)
  assert E.main(['--model_dir', str(tmp_path), '--enbl_fake_prune', '--fake_prune_ratio', '0.3']) == 0
  out = E.load_exported(str(tmp_path / 'model_shrunk.npz'))
  assert out['model/conv/kernel'].shape == (3, 3, 7, 6) and len(out['model/conv/kernel/gather']) == 7


# ---------------------------------------------------------------------------------------------------------------------
# integer export of the UniformQuantLearner (pocketflow_amd/tools/conversion/export_quant_int8_model.py)
# ---------------------------------------------------------------------------------------------------------------------
import pytest


@pytest.mark.parametrize('bits', [1, 2, 3, 4, 8])
@pytest.mark.parametrize('mode,shape,bucket_size', [('tensor', (3, 3, 5, 7), 0), ('channel', (3, 3, 5, 7), 0), ('split', (3, 3, 5, 7), 64),
                                                    ('split', (1, 1, 33, 10), 256), ('channel', (40, 12), 0)])
def test_int_codes_reproduce_the_reference_quantiser_bit_for_bit(bits, mode, shape, bucket_size):
  """encode / decode against oracle/pf_oracle.py uniform_quantize (= the reference's __uniform_quantize executed, see
  tests/test_oracle_golden.py): the decoded tensor IS the fake-quantised tensor, for every bucket mode incl. the padded last
  split bucket; codes span [0, 2^bits - 1]; the packed size is bits per weight."""
  from oracle import pf_oracle as O
  from pocketflow_amd.tools.conversion import export_quant_int8_model as E
  rng = np.random.RandomState(bits * 100 + len(shape) + bucket_size)
  w = (rng.randn(*shape) * 0.1).astype(np.float32)
  wq, info = O.uniform_quantize(w, bits, 'weight', mode != 'tensor', 'split' if mode == 'split' else 'channel', bucket_size or 256)
  alpha, beta = np.atleast_1d(np.asarray(info['alpha'], np.float32)).reshape(-1), np.atleast_1d(np.asarray(info['beta'], np.float32)).reshape(-1)
  m = {'tensor': E.MODE_TENSOR, 'channel': E.MODE_CHANNEL, 'split': E.MODE_SPLIT}[mode]
  packed, meta = E.encode_tensor(wq, alpha, beta, bits, m, bucket_size)
  assert packed.nbytes == -(-w.size * bits // 8)
  back = E.decode_tensor(packed, alpha, beta, meta)
  assert np.array_equal(back.view(np.uint32), wq.view(np.uint32))
  codes = E.unpack_codes(packed, bits, w.size)
  assert codes.min() == 0 and codes.max() == 2 ** bits - 1       # every bucket holds its own min and max


def test_int_export_rejects_what_it_cannot_represent():
  from pocketflow_amd.tools.conversion import export_quant_int8_model as E
  w = np.linspace(-1, 1, 64, dtype=np.float32).reshape(1, 1, 8, 8)
  with pytest.raises(ValueError):                                # not on the grid of this (alpha, beta)
    E.encode_tensor(w, np.array([2.0], np.float32), np.array([-1.0], np.float32), 2, E.MODE_TENSOR, 0)
  with pytest.raises(ValueError):
    E.encode_tensor(w, np.array([2.0], np.float32), np.array([-1.0], np.float32), 9, E.MODE_TENSOR, 0)
