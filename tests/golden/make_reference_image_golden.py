#!/usr/bin/env python
"""Generate tests/golden/reference_image.npz by EXECUTING THE REFERENCE'S OWN input-pipeline functions
(utils/external/imagenet_preprocessing.py: preprocess_image and everything it calls; datasets/ilsvrc12_dataset.py:
parse_example_proto is covered by tests/test_tfrecord.py against the protobuf runtime).  Same method and caveats as
make_reference_golden.py; tf.image ops are the NumPy / Pillow stand-ins of oracle/tf_stub.py, with the random crop
window and flip injected (`tf_stub.image_hooks`).  Build container only.

    python tests/golden/make_reference_image_golden.py
"""
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_reference_golden as G  # noqa: E402

tf, T, stub = G.tf, G.tf_stub.T, G.tf_stub


def synthetic_jpeg(rng, h, w, quality=90):
  """A smooth random image (so that JPEG artefacts stay small) as baseline-JPEG bytes."""
  from PIL import Image
  yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
  img = np.zeros((h, w, 3))
  for c in range(3):
    for _ in range(4):
      fy, fx, ph = rng.uniform(0.01, 0.15), rng.uniform(0.01, 0.15), rng.uniform(0, 6.28)
      img[:, :, c] += rng.uniform(20, 60) * np.sin(fy * yy + fx * xx + ph)
    img[:, :, c] += rng.uniform(90, 160)
  buf = io.BytesIO()
  Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, format='JPEG', quality=quality)
  return buf.getvalue()


def main():
  src = open(os.path.join(G.REF, 'utils/external/imagenet_preprocessing.py')).read()
  ns = {'tf': tf, '__name__': 'lifted'}
  exec(compile(src, os.path.join(G.REF, 'utils/external/imagenet_preprocessing.py'), 'exec'), ns)   # the whole module
  rng = np.random.RandomState(77)
  arrays, cases = {}, []
  for name, (h, w) in (('landscape', (300, 451)), ('portrait', (517, 260)), ('small', (97, 131)), ('square', (256, 256))):
    jpeg = synthetic_jpeg(rng, h, w)
    arrays['%s/jpeg' % name] = np.frombuffer(jpeg, dtype=np.uint8)
    size = 112 if name == 'landscape' else 56            # one case at the real output size, the rest small (fixture size)
    out = ns['preprocess_image'](jpeg, None, size, size, 3, is_training=False)
    arrays['%s/eval' % name] = out.numpy()
    for k, (window, flip) in enumerate((((0, 0, h, w), False), ((h // 7, w // 5, h // 2, (2 * w) // 3), True),
                                        ((h // 3, 3, h // 3 + 11, w // 2 - 5), False))):
      stub.image_hooks.update(window=window, flip=flip)
      tsize = 112 if (name == 'landscape' and k == 1) else 56
      out = ns['preprocess_image'](jpeg, T(np.zeros((1, 0, 4), np.float32)), tsize, tsize, 3, is_training=True)
      arrays['%s/train%d' % (name, k)] = out.numpy()
      cases.append(dict(name=name, k=k, window=list(window), flip=flip, out=tsize))
    nh, nw = ns['_smallest_size_at_least'](T(np.int32(h)), T(np.int32(w)), 256)
    cases.append(dict(name=name, size=[h, w], resized=[int(nh.a), int(nw.a)], eval_out=size))
  for (h, w) in ((480, 640), (333, 500), (1200, 900), (256, 256), (255, 257), (3000, 17)):
    nh, nw = ns['_smallest_size_at_least'](T(np.int32(h)), T(np.int32(w)), 256)
    cases.append(dict(size=[h, w], resized=[int(nh.a), int(nw.a)]))
  # CIFAR-10: datasets/cifar10_dataset.py parse_fn on hand-made records (crop offsets and flip injected)
  cns = G.lift('datasets/cifar10_dataset.py', ['parse_fn'],
               {'LABEL_BYTES': 1, 'IMAGE_HEI': 32, 'IMAGE_WID': 32, 'IMAGE_CHN': 3, 'IMAGE_BYTES': 3072,
                'IMAGE_AVE': tf.constant([[[125.3, 123.0, 113.9]]], dtype=tf.float32),
                'IMAGE_STD': tf.constant([[[63.0, 62.1, 66.7]]], dtype=tf.float32)})
  G.FLAGS.nb_classes = 10
  crng = np.random.RandomState(5)
  records = crng.randint(0, 256, size=(4, 3073)).astype(np.uint8)
  records[:, 0] = [3, 0, 9, 7]
  arrays['cifar/records'] = records
  for i, rec in enumerate(records):
    img, lab = cns['parse_fn'](rec.tobytes(), False)
    arrays['cifar/eval%d' % i], arrays['cifar/label%d' % i] = img.numpy(), lab.numpy()
    for k, (oy, ox, flip) in enumerate(((0, 0, False), (8, 8, True), (3, 6, True), (4, 4, False))):
      stub.image_hooks.update(crop=(oy, ox), flip=flip)
      img, _ = cns['parse_fn'](rec.tobytes(), True)
      arrays['cifar/train%d_%d' % (i, k)] = img.numpy()
  cases.append(dict(cifar_augment=[[0, 0, False], [8, 8, True], [3, 6, True], [4, 4, False]]))
  np.savez_compressed(os.path.join(HERE, 'reference_image.npz'), **arrays)
  with open(os.path.join(HERE, 'reference_image.json'), 'w') as f:
    json.dump({'cases': cases, 'means': [ns['_R_MEAN'], ns['_G_MEAN'], ns['_B_MEAN']], 'resize_min': ns['_RESIZE_MIN']}, f, indent=1)
  print('wrote %d arrays (%.1f KiB)' % (len(arrays), os.path.getsize(os.path.join(HERE, 'reference_image.npz')) / 1024.0))


if __name__ == '__main__':
  main()
