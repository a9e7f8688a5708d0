#!/usr/bin/env python
"""Reference-executed fixture for SURVEY section 8a row a17 (feature extraction of the channel pruner).

Runs HERE (needs /root/reference); the GPU box only reads the committed output (reference_cp_features.npz / .json).

The reference's own methods are lifted out of /root/reference with `ast` (no source is copied) and EXECUTED over
oracle/tf_cp_graph_stub.py, a NumPy graph with the TF Graph / Operation / Tensor / Session surface they touch:

  learners/channel_pruning/channel_pruner.py  ChannelPruner.__extract_output_of_conv_and_sum (:215-227), extract_features
                                              (:263-341), __create_extractor (:343-359), __extract_new_features (:361-389),
                                              __extract_input (:391-412), residual_branch_diff (:579-586)
  learners/channel_pruning/model_wrapper.py   Model.get_operations_by_type(s), get_outputs_by_type(s), get_output_by_op,
                                              get_input_by_op, get_names, get_conv_def, get_outname_by_opname, output_height /
                                              width / channels, get_Add_if_is_last_in_resblock (:60-135, 195-254, 304-341)

Stand-ins: `Model.param_shape` (the reference goes through slim.get_variables_by_name) reads the kernel shape off the stand-in
graph; `tf.extract_image_patches` is the NumPy function the oracle uses too; np.random is seeded (the reference samples with the
global, unseeded generator).  Recipe (network seed, image seed, sampling seed, pruned channels) is stored beside the arrays;
tests/test_cp_features_oracle.py rebuilds everything from it and requires oracle/cp_features_oracle.py to match bit for bit."""
import contextlib
import json
import os
import sys
import types
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_reference_golden as G  # noqa: E402  (lift(), the tf stub with its FLAGS)
from oracle import tf_cp_graph_stub as S  # noqa: E402
from oracle.cp_features_oracle import extract_image_patches  # noqa: E402

RECIPE = {'net_seed': 5, 'batch': 3, 'hw': 12, 'image_seed': 11, 'sample_seed': 123, 'nb_batches': 2, 'nb_points': 4,
          # input channels zeroed after the features were extracted (the "pruning done so far")
          'pruned': {'b1/conv1/Conv2D': [1, 5], 'b1/conv3/Conv2D': [0], 'b2/conv2/Conv2D': [2, 3], 'b1/proj/Conv2D': [7]}}


def images_of(recipe):
  rng = np.random.RandomState(recipe['image_seed'])
  return [rng.randn(recipe['batch'], recipe['hw'], recipe['hw'], 3).astype(np.float32) for _ in range(recipe['nb_batches'])]


def apply_pruning(net, recipe):
  for name, chans in recipe['pruned'].items():
    net.kernels[name].weight[:, :, chans, :] = 0.0


def main():
  FLAGS = G.FLAGS
  net = S.build_standin(RECIPE['net_seed'], RECIPE['batch'], RECIPE['hw'])
  sess = S.Session(net.g)

  def _patches(inp, ksizes, strides, rates, padding):
    assert list(rates) == [1, 1, 1, 1]
    fn = lambda v: extract_image_patches(v, ksizes[1], ksizes[2], strides[1], strides[2], padding)
    return S.Operation(net.g, inp.op.name + '/ExtractImagePatches', 'ExtractImagePatches', [inp], fn, [None] * 4).outputs[0]

  log = types.SimpleNamespace(info=lambda *a, **k: None, debug=lambda *a, **k: None, error=lambda *a, **k: None)
  tfx = types.SimpleNamespace(logging=log, extract_image_patches=_patches,
                              variable_scope=lambda *a, **k: contextlib.nullcontext())
  slim = types.SimpleNamespace(queues=types.SimpleNamespace(QueueRunners=lambda sess: contextlib.nullcontext()))
  mw = G.lift('learners/channel_pruning/model_wrapper.py',
              ['Model.get_operations_by_type', 'Model.get_operations_by_types', 'Model.get_outputs_by_type',
               'Model.get_outputs_by_types', 'Model.get_output_by_op', 'Model.get_input_by_op', 'Model.get_names',
               'Model.get_conv_def', 'Model.get_outname_by_opname', 'Model.output_height', 'Model.output_width',
               'Model.output_channels', 'Model.get_Add_if_is_last_in_resblock'], {'tf': tfx, 'slim': slim})
  Model = mw['Model']
  for m in ('get_names', 'get_outname_by_opname', 'get_Add_if_is_last_in_resblock'):      # lift() strips decorators
    setattr(Model, m, classmethod(getattr(Model, m)))
  Model.param_shape = lambda self, op: [d.value for d in op.inputs[1].shape]
  model = Model.__new__(Model)
  model.sess, model.g, model.data_format = sess, net.g, 'NHWC'

  cp = G.lift('learners/channel_pruning/channel_pruner.py',
              ['ChannelPruner.__extract_output_of_conv_and_sum', 'ChannelPruner.extract_features',
               'ChannelPruner.__create_extractor', 'ChannelPruner.__extract_new_features', 'ChannelPruner.__extract_input',
               'ChannelPruner.residual_branch_diff'], {'tf': tfx, 'slim': slim, 'OrderedDict': OrderedDict})
  CP = cp['ChannelPruner']
  pr = CP.__new__(CP)
  pr._model, pr.data_format, pr.mem_images = model, 'NHWC', net.mem_images
  batches = images_of(RECIPE)
  pr.images = S.Queue(batches)
  pr.labels = S.Queue([np.zeros(RECIPE['batch'], np.int64)] * RECIPE['nb_batches'])
  FLAGS.cp_nb_points_per_layer, FLAGS.batch_size, FLAGS.cp_nb_batches = RECIPE['nb_points'], RECIPE['batch'], RECIPE['nb_batches']
  pr._ChannelPruner__extract_output_of_conv_and_sum()
  pr._ChannelPruner__create_extractor()
  np.random.seed(RECIPE['sample_seed'])
  pr.extract_features()

  out, meta = {}, {'recipe': RECIPE, 'names': list(pr.names), 'unique_names': list(pr.feats_dict.keys())}
  for name, f in pr.feats_dict.items():
    out['feats/' + name] = f
  for key, v in pr.points_dict.items():
    if isinstance(key, tuple) and len(key) == 3:
      out['points/%d/%s/%s' % key] = v
  meta['nb_points_per_batch'] = int(pr.points_dict['nb_points_per_batch'])

  apply_pruning(net, RECIPE)
  convs = model.get_operations_by_type()
  meta['convs'] = [op.name for op in convs]
  meta['adds'] = {}
  for op in convs:
    out['input/' + op.name] = pr._ChannelPruner__extract_input(op)
    add = model.get_Add_if_is_last_in_resblock(op)
    if add is not None:
      meta['adds'][op.name] = add.name
      out['diff/' + add.name] = pr.residual_branch_diff(add.name)
  np.savez_compressed(os.path.join(HERE, 'reference_cp_features.npz'), **out)
  with open(os.path.join(HERE, 'reference_cp_features.json'), 'w') as f:
    json.dump(meta, f, indent=1, sort_keys=True)
  print('wrote %d arrays; names = %s; adds = %s' % (len(out), meta['names'], meta['adds']))


if __name__ == '__main__':
  main()
