"""oracle/cp_features_oracle.py against the arrays the reference's OWN sampling code produced
(tests/golden/make_reference_cp_features_golden.py executes it over oracle/tf_cp_graph_stub.py) -- bit for bit, from the
same seeded recipe; plus known answers for the restated tf.extract_image_patches."""
import json
import os

import numpy as np
import pytest

from oracle import cp_features_oracle as CF
from oracle import tf_cp_graph_stub as S

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def golden():
  arrays = np.load(os.path.join(HERE, 'golden', 'reference_cp_features.npz'))
  with open(os.path.join(HERE, 'golden', 'reference_cp_features.json')) as f:
    meta = json.load(f)
  return arrays, meta


def _standin(recipe):
  net = S.build_standin(recipe['net_seed'], recipe['batch'], recipe['hw'])
  sess = S.Session(net.g)
  run = lambda images, names: sess.run(list(names), feed_dict={net.mem_images: images})
  rng = np.random.RandomState(recipe['image_seed'])
  batches = [rng.randn(recipe['batch'], recipe['hw'], recipe['hw'], 3).astype(np.float32) for _ in range(recipe['nb_batches'])]
  shapes = {o.outputs[0].name: tuple(d.value for d in o.outputs[0].shape[1:]) for o in net.g.operations
            if o.type in ('Conv2D', 'Add')}
  return net, run, batches, shapes


def test_names_of_convolutions_and_their_residual_sums(golden):
  __, meta = golden
  net = S.build_standin(meta['recipe']['net_seed'], meta['recipe']['batch'], meta['recipe']['hw'])
  ops = net.g.ops()
  assert CF.conv_add_names(ops) == meta['names']
  adds = {o[0]: CF.add_if_is_last_in_resblock(ops, o[0]) for o in ops if o[1] == 'Conv2D'}
  assert {k: v for k, v in adds.items() if v is not None} == meta['adds']
  # the projection shortcut closes the block as well (its only consumer is the Add); a convolution whose output reaches another
  # convolution through BN / ReLU6 / depthwise does not
  assert adds['b1/proj/Conv2D'] == 'b1/add:0' and adds['tail/pw0/Conv2D'] is None and adds['b1/conv2/Conv2D'] is None


def test_sampled_features_inputs_and_residual_diffs_equal_the_reference_code(golden):
  arrays, meta = golden
  r = meta['recipe']
  net, run, batches, shapes = _standin(r)
  feats, points = CF.extract_features(run, CF.conv_add_names(net.g.ops()), shapes, batches, r['nb_points'],
                                      np.random.RandomState(r['sample_seed']))
  assert list(feats.keys()) == meta['unique_names'] and points['nb_points_per_batch'] == meta['nb_points_per_batch']
  for name in meta['unique_names']:
    assert feats[name].dtype == np.float64
    np.testing.assert_array_equal(feats[name], arrays['feats/' + name], err_msg=name)
    for b in range(r['nb_batches']):
      for k in ('x_samples', 'y_samples'):
        np.testing.assert_array_equal(points[(b, name, k)], arrays['points/%d/%s/%s' % (b, name, k)])
  # the pruning done so far, then the inputs of every convolution and the residual diffs of the CURRENT network
  for name, chans in r['pruned'].items():
    net.kernels[name].weight[:, :, chans, :] = 0.0
  by_name = {o.name: o for o in net.g.operations}
  for cname in meta['convs']:
    op = by_name[cname]
    kh, kw, c, __ = [d.value for d in op.inputs[1].shape]
    st = op.get_attr('strides')
    conv = dict(name=cname, input=op.inputs[0].name, h=kh, w=kw, c=c, strides=(st[1], st[2]), padding=op.get_attr('padding'))
    X = CF.extract_input(run, conv, points, r['nb_batches'])
    np.testing.assert_array_equal(X, arrays['input/' + cname], err_msg=cname)
  for cname, add in meta['adds'].items():
    d = CF.residual_branch_diff(run, add, shapes, points, r['nb_batches'], feats)
    np.testing.assert_array_equal(d, arrays['diff/' + add], err_msg=add)
    assert np.abs(d).max() > 0          # the pruning did change the sums


def test_extract_image_patches_known_answers():
  x = np.arange(2 * 4 * 5 * 2, dtype=np.float32).reshape(2, 4, 5, 2)
  # 1x1 stride 2, VALID: plain subsampling
  np.testing.assert_array_equal(CF.extract_image_patches(x, 1, 1, 2, 2, 'VALID'), x[:, ::2, ::2, :])
  # 3x3 stride 1 SAME: centre tap is the input, the window of pixel (0, 0) starts one row / column outside (zeros)
  p = CF.extract_image_patches(x, 3, 3, 1, 1, 'SAME')
  assert p.shape == (2, 4, 5, 18)
  np.testing.assert_array_equal(p[..., 8:10], x)
  assert (p[:, 0, :, 0:6] == 0).all() and (p[:, :, 0, [0, 1, 6, 7, 12, 13]] == 0).all()
  np.testing.assert_array_equal(p[:, 1, 1, 0:2], x[:, 0, 0, :])
  # 3x3 stride 2 SAME on an even size: TF pads 0 before and 1 after (total//2 before)
  p = CF.extract_image_patches(x[:, :, :4], 3, 3, 2, 2, 'SAME')
  assert p.shape == (2, 2, 2, 18)
  np.testing.assert_array_equal(p[:, 0, 0, 0:2], x[:, 0, 0, :])
  assert (p[:, 1, :, 12:18] == 0).all()
