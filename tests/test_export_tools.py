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
