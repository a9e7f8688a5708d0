"""ILSVRC-12 (reference datasets/ilsvrc12_dataset.py:27-130).

With `--data_dir_local <dir holding train-*-of-* / validation-*-of-* TFRecord shards>` the shards are streamed
(datasets/tfrecord_image_iterator.py): tf.train.Example -> JPEG decode + crop window on host threads -> one device
kernel for resize / flip / central crop / mean subtraction (utils/external/imagenet_preprocessing.py:226-260).
Without a data directory: a seeded synthetic stream with the same tensor contract (the benchmark configurations;
SURVEY section 8d) -- 224x224x3, channel means 123.68 / 116.78 / 103.94 subtracted, NO division by a standard
deviation, 1001 classes (slim's background class)."""
from __future__ import annotations

import glob
import os

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
    if FLAGS.data_disk not in ('local', 'hdfs'):
      raise ValueError('unrecognized data disk: ' + str(FLAGS.data_disk))
    if FLAGS.data_disk == 'hdfs':
      raise ValueError('HDFS input is outside the MI355X hot path (SURVEY section 2, row 24)')
    self.files = []
    if FLAGS.data_dir_local:
      pattern = 'train-*-of-*' if is_train else 'validation-*-of-*'
      self.files = sorted(glob.glob(os.path.join(FLAGS.data_dir_local, pattern)))

  def build(self, enbl_trn_val_split=False, device=None):
    if not self.files:
      return super(Ilsvrc12Dataset, self).build(enbl_trn_val_split, device)
    from pocketflow_amd.datasets.tfrecord_image_iterator import TFRecordImageIterator
    from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw
    rank, size = (mgw.rank(), mgw.size()) if FLAGS.enbl_multi_gpu else (0, 1)
    files = self.files
    if self.is_train and size > 1:
      files = files[rank::size] or files               # filenames.shard(size, rank), abstract_dataset.py:80-81
    seed = FLAGS.synthetic_seed + rank + (0 if self.is_train else 100003)
    device = device if device is not None else self.device
    hw = (FLAGS.image_size, FLAGS.image_size)
    # the resize kernel writes the compute dtype directly (bf16 = round-to-nearest-even of the float32 result, i.e.
    # what the learners' cast would produce): the float32 batch of the reference's contract never exists in HBM
    import torch
    dtype = torch.bfloat16 if ('compute_dtype' in FLAGS and FLAGS.compute_dtype == 'bfloat16') else torch.float32

    def make(skip=0, take=None, seed_off=0):
      return TFRecordImageIterator(files, self.batch_size, self.is_train, FLAGS.nb_classes, seed + seed_off, device, hw,
                                   skip=skip, take=take, cycle_length=FLAGS.cycle_length, nb_threads=FLAGS.nb_threads,
                                   buffer_size=FLAGS.buffer_size, prefetch_size=FLAGS.prefetch_size, dtype=dtype)
    if self.is_train and enbl_trn_val_split:
      return make(skip=FLAGS.nb_smpls_val), make(take=FLAGS.nb_smpls_val, seed_off=50021)   # dataset.skip / dataset.take
    return make()

  def make_batch(self, rng, batch_size):
    h = w = FLAGS.image_size
    raw = rng.randint(0, 256, size=(batch_size, h, w, IMAGE_CHN)).astype(np.float32)
    images = (raw - _MEAN).astype(np.float32)
    cls = rng.randint(0, FLAGS.nb_classes, size=(batch_size,))
    labels = np.zeros((batch_size, FLAGS.nb_classes), dtype=np.float32)
    labels[np.arange(batch_size), cls] = 1.0
    return images, labels
