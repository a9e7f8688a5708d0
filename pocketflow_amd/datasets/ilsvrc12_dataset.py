"""ILSVRC-12 (synthetic).  Flags, shapes and value range of datasets/ilsvrc12_dataset.py:27-93 and
utils/external/imagenet_preprocessing.py:40-43,260: 224x224x3, channel means 123.68 / 116.78 /
103.94 subtracted, NO division by a standard deviation, 1001 classes (slim's background class)."""
from __future__ import annotations

import numpy as np

from pocketflow_amd.datasets.abstract_dataset import AbstractDataset
from pocketflow_amd.flags import FLAGS, flags

flags.DEFINE_integer('nb_classes', 1001, '# of classes')
flags.DEFINE_integer('nb_smpls_train', 1281167, '# of samples for training')
flags.DEFINE_integer('nb_smpls_val', 10000, '# of samples for validation')
flags.DEFINE_integer('nb_smpls_eval', 50000, '# of samples for evaluation')
flags.DEFINE_integer('batch_size', 64, 'batch size per GPU for training')
flags.DEFINE_integer('batch_size_eval', 100, 'batch size for evaluation')
flags.DEFINE_integer('image_size', 224, 'synthetic image height / width (224 in the reference)')

IMAGE_CHN = 3
_MEAN = np.array([123.68, 116.78, 103.94], dtype=np.float32)


class Ilsvrc12Dataset(AbstractDataset):
  def __init__(self, is_train):
    super(Ilsvrc12Dataset, self).__init__(is_train)
    self.batch_size = FLAGS.batch_size if is_train else FLAGS.batch_size_eval
    self.image_shape = (FLAGS.image_size, FLAGS.image_size, IMAGE_CHN)

  def make_batch(self, rng, batch_size):
    h = w = FLAGS.image_size
    raw = rng.randint(0, 256, size=(batch_size, h, w, IMAGE_CHN)).astype(np.float32)
    images = (raw - _MEAN).astype(np.float32)
    cls = rng.randint(0, FLAGS.nb_classes, size=(batch_size,))
    labels = np.zeros((batch_size, FLAGS.nb_classes), dtype=np.float32)
    labels[np.arange(batch_size), cls] = 1.0
    return images, labels
