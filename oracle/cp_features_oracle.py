"""TEST INFRASTRUCTURE (only tests/ may import this): NumPy restatement of the feature sampling of the reference's channel
pruner, SURVEY section 8a row a17 -- /root/reference/learners/channel_pruning/channel_pruner.py

    __extract_output_of_conv_and_sum   :215-227      conv_add_names
    extract_features                   :263-341      extract_features
    __create_extractor                 :343-359      (tf.extract_image_patches -> extract_image_patches)
    __extract_new_features             :361-389      extract_new_features
    __extract_input                    :391-412      extract_input
    residual_branch_diff               :579-586      residual_branch_diff
  and of learners/channel_pruning/model_wrapper.py
    get_Add_if_is_last_in_resblock     :304-341      add_if_is_last_in_resblock

The network is not part of this file: the caller supplies `ops`, the graph as an ordered list of (name, type, [input names])
in creation order (what `g.get_operations()` walks), and `run(images, names) -> [NHWC arrays]`, the stand-in for
`sess.run(names, feed_dict={mem_images: images})`.  Tensor names are TF's: '<op name>:0'.

Pinned: tests/golden/make_reference_cp_features_golden.py EXECUTES the reference's own methods (lifted with `ast`) over a small
NumPy graph stand-in and stores what they return; tests/test_cp_features_oracle.py requires this file to reproduce those
arrays bit for bit from the same seeded recipe.  What stays a restatement is `tf.extract_image_patches` itself (SAME / VALID
windows in (row, column, depth) order), which exists on both sides of that comparison as the same NumPy function.

The reference draws its sample points from the GLOBAL, unseeded `np.random`; here the generator is an argument (a
`np.random.RandomState`), consumed in the reference's order: per batch, per name in `conv_add_names` order, rows then columns.
"""
from collections import OrderedDict

import numpy as np

# op types the walk from a convolution to "its" Add may pass through (model_wrapper.py:322-327)
_PASS_THROUGH = ('Relu', 'FusedBatchNorm', 'DepthwiseConv2dNative', 'MaxPool', 'Relu6')


def _consumers(ops, name):
  """Ops reading the first output of op `name`, in creation order (Tensor.consumers())."""
  return [o for o in ops if name in o[2]]


def add_if_is_last_in_resblock(ops, conv_name):
  """Name of the Add OUTPUT TENSOR if the convolution is the last one before the sum of a residual block, else None
  (model_wrapper.py:304-341).  The walk follows the FIRST pass-through consumer of each tensor and stops at the LAST consumer
  examined otherwise -- exactly the reference's loop, including its reuse of the loop variable."""
  by_name = {o[0]: o for o in ops}
  curr = by_name[conv_name]
  while True:
    nxt = _consumers(ops, curr[0])
    go_on = False
    for curr in nxt:                      # `curr_op` is rebound by the loop, as in the reference
      if curr[1] in _PASS_THROUGH:
        go_on = True
        break
    if go_on:
      continue
    if curr[1] == 'Add':
      return curr[0] + ':0'
    return None


def conv_add_names(ops):
  """Output names of every Conv2D, each followed by the Add it closes, duplicates kept (:215-227); `extract_features`
  removes the duplicates while keeping first positions (:299)."""
  names = []
  for o in ops:
    if o[1] != 'Conv2D':
      continue
    names.append(o[0] + ':0')
    add = add_if_is_last_in_resblock(ops, o[0])
    if add is not None:
      names.append(add)
  return names


def extract_features(run, names, shapes, batches, nb_points_per_layer, rng):
  """:263-341.  `shapes[name]` = (H, W, C) of the tensor, `batches` = list of image arrays.  Returns (feats_dict,
  points_dict) with the reference's keys: (batch, 0) -> images, (batch, name, 'x_samples' | 'y_samples')."""
  names = list(OrderedDict.fromkeys(names))
  batch_size = batches[0].shape[0]
  per_batch = nb_points_per_layer * batch_size
  points = {'nb_points_per_batch': per_batch}
  feats = {n: np.ndarray(shape=(per_batch * len(batches), shapes[n][2])) for n in names}
  idx = 0
  for b, data in enumerate(batches):
    points[(b, 0)] = data
    outs = run(data, names)
    for feat, name in zip(outs, names):
      xs = rng.randint(0, shapes[name][0] - 0, nb_points_per_layer)
      ys = rng.randint(0, shapes[name][1] - 0, nb_points_per_layer)
      points[(b, name, 'x_samples')] = xs.copy()
      points[(b, name, 'y_samples')] = ys.copy()
      feats[name][idx:idx + per_batch] = feat[:, xs, ys, :].reshape((per_batch, -1))
    idx += per_batch
  return feats, points


def extract_new_features(run, names, shapes, points, nb_batches):
  """:361-389: the CURRENT model at the stored points of `names`."""
  per_batch = points['nb_points_per_batch']
  feats = {n: np.ndarray(shape=(per_batch * nb_batches, shapes[n][2])) for n in names}
  idx = 0
  for b in range(nb_batches):
    outs = run(points[(b, 0)], names)
    for feat, name in zip(outs, names):
      xs, ys = points[(b, name, 'x_samples')], points[(b, name, 'y_samples')]
      feats[name][idx:idx + per_batch] = feat[:, xs, ys, :].reshape((per_batch, -1))
    idx += per_batch
  return feats


def residual_branch_diff(run, sum_name, shapes, points, nb_batches, feats_dict):
  """:579-586: original minus current value of the residual sum, at the SUM's own sample points."""
  new = extract_new_features(run, [sum_name], shapes, points, nb_batches)
  return feats_dict[sum_name] - new[sum_name]


def extract_image_patches(x, kh, kw, sh, sw, padding):
  """tf.extract_image_patches(x NHWC, ksizes [1,kh,kw,1], strides [1,sh,sw,1], rates 1): [B, Ho, Wo, kh*kw*C], window
  elements in (row, column, depth) order; SAME pads total//2 before, the rest after, with zeros."""
  B, H, W, C = x.shape
  if padding in ('SAME', b'SAME'):
    ho, wo = -(-H // sh), -(-W // sw)
    th, tw = max((ho - 1) * sh + kh - H, 0), max((wo - 1) * sw + kw - W, 0)
    x = np.pad(x, ((0, 0), (th // 2, th - th // 2), (tw // 2, tw - tw // 2), (0, 0)))
  else:
    ho, wo = (H - kh) // sh + 1, (W - kw) // sw + 1
  out = np.zeros((B, ho, wo, kh * kw * C), x.dtype)
  for r in range(kh):
    for s in range(kw):
      out[:, :, :, (r * kw + s) * C:(r * kw + s + 1) * C] = x[:, r:r + (ho - 1) * sh + 1:sh, s:s + (wo - 1) * sw + 1:sw, :]
  return out


def extract_input(run, conv, points, nb_batches):
  """:391-412.  `conv` = dict(name, input (tensor name), h, w, c, strides (sh, sw), padding): patches of the convolution's
  input at the sample points of its OUTPUT, as [n, h, w, c]."""
  out_name = conv['name'] + ':0'
  Xs = []
  for b in range(nb_batches):
    inp = run(points[(b, 0)], [conv['input']])[0]
    feat = extract_image_patches(inp, conv['h'], conv['w'], conv['strides'][0], conv['strides'][1], conv['padding'])
    xs, ys = points[(b, out_name, 'x_samples')], points[(b, out_name, 'y_samples')]
    X = feat[:, xs, ys, :].reshape((-1, feat.shape[-1]))
    Xs.append(X.reshape((X.shape[0], conv['h'], conv['w'], conv['c'])))
  return np.vstack(Xs)
