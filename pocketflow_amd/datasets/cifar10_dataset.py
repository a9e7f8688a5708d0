"""CIFAR-10 (reference datasets/cifar10_dataset.py:26-110).

With `--data_dir_local <dir holding data_batch_*.bin / test_batch.bin>` the binary records are read and the
reference's parse_fn is applied: record = 1 label byte + 3x32x32 CHW image bytes -> HWC float32,
(x - [125.3, 123.0, 113.9]) / [63.0, 62.1, 66.7]; training adds zero-padding to 40x40 (AFTER the
standardisation, as resize_image_with_crop_or_pad does), a random 32x32 crop and a random horizontal flip
(:43-70).  Without a data directory a seeded synthetic stream with the same tensor contract is produced
(the benchmark configurations).
"""
from __future__ import annotations

import glob
import os

import numpy as np
import torch

from pocketflow_amd.datasets.abstract_dataset import AbstractDataset
from pocketflow_amd.datasets.record_iterator import RecordIterator
from pocketflow_amd.flags import FLAGS, flags

flags.DEFINE_integer('nb_classes', 10, '# of classes')
flags.DEFINE_integer('nb_smpls_train', 50000, '# of samples for training')
flags.DEFINE_integer('nb_smpls_val', 5000, '# of samples for validation')
flags.DEFINE_integer('nb_smpls_eval', 10000, '# of samples for evaluation')
flags.DEFINE_integer('batch_size', 128, 'batch size per GPU for training')
flags.DEFINE_integer('batch_size_eval', 100, 'batch size for evaluation')

LABEL_BYTES = 1
IMAGE_HEI, IMAGE_WID, IMAGE_CHN = 32, 32, 3
IMAGE_BYTES = IMAGE_CHN * IMAGE_HEI * IMAGE_WID
RECORD_BYTES = LABEL_BYTES + IMAGE_BYTES
_MEAN = np.array([125.3, 123.0, 113.9], dtype=np.float32)
_STD = np.array([63.0, 62.1, 66.7], dtype=np.float32)


def read_records(paths):
  """(images uint8 [N, 32, 32, 3], labels int64 [N]) of CIFAR-10 binary files (FixedLengthRecordDataset)."""
  raw = np.concatenate([np.fromfile(p, dtype=np.uint8) for p in paths])
  if raw.size % RECORD_BYTES:
    raise ValueError('CIFAR-10 binary files must hold %d-byte records' % RECORD_BYTES)
  rec = raw.reshape(-1, RECORD_BYTES)
  labels = rec[:, 0].astype(np.int64)
  images = rec[:, LABEL_BYTES:].reshape(-1, IMAGE_CHN, IMAGE_HEI, IMAGE_WID).transpose(0, 2, 3, 1)
  return np.ascontiguousarray(images), labels


def standardize(u8: torch.Tensor) -> torch.Tensor:
  """decode_raw -> HWC float32 -> (x - IMAGE_AVE) / IMAGE_STD (reference :58-62)."""
  dev = u8.device
  return (u8.to(torch.float32) - torch.from_numpy(_MEAN).to(dev)) / torch.from_numpy(_STD).to(dev)


def augment(x: torch.Tensor, oy: torch.Tensor, ox: torch.Tensor, flip: torch.Tensor) -> torch.Tensor:
  """resize_image_with_crop_or_pad(+8) -> crop at (oy, ox) in [0, 8] -> optional left/right flip, batched (:65-68).
  The padding is zeros in the STANDARDISED domain (the reference pads after the standardisation)."""
  dev, B = x.device, x.shape[0]
  xp = torch.nn.functional.pad(x, (0, 0, 4, 4, 4, 4))                       # [B, 40, 40, 3]
  ar = torch.arange(IMAGE_HEI, device=dev)
  rows = (oy.to(dev)[:, None] + ar[None, :])                                # [B, 32]
  cols = (ox.to(dev)[:, None] + ar[None, :])
  cols = torch.where(flip.to(dev)[:, None], cols.flip(1), cols)             # flip = reversed column order
  bidx = torch.arange(B, device=dev)[:, None, None]
  return xp[bidx, rows[:, :, None], cols[:, None, :], :]


def make_transform(is_train: bool):
  """parse_fn after decode_raw: standardise; training: pad 4 (zeros) -> random crop 32x32 -> random flip."""
  def transform(u8: torch.Tensor, gen: torch.Generator) -> torch.Tensor:
    x = standardize(u8)
    if not is_train:
      return x
    B = x.shape[0]
    oy = torch.randint(0, 9, (B,), generator=gen)
    ox = torch.randint(0, 9, (B,), generator=gen)
    flip = torch.rand(B, generator=gen) < 0.5
    return augment(x, oy, ox, flip)
  return transform


class Cifar10Dataset(AbstractDataset):
  def __init__(self, is_train):
    super(Cifar10Dataset, self).__init__(is_train)
    self.batch_size = FLAGS.batch_size if is_train else FLAGS.batch_size_eval
    self.image_shape = (IMAGE_HEI, IMAGE_WID, IMAGE_CHN)
    if FLAGS.data_disk not in ('local', 'hdfs'):
      raise ValueError('unrecognized data disk: ' + str(FLAGS.data_disk))
    if FLAGS.data_disk == 'hdfs':
      raise ValueError('HDFS input is outside the MI355X hot path (SURVEY section 2, row 24)')
    self.files = []
    if FLAGS.data_dir_local:
      pattern = 'data_batch_*.bin' if is_train else 'test_batch.bin'
      self.files = sorted(glob.glob(os.path.join(FLAGS.data_dir_local, pattern)))

  def make_batch(self, rng, batch_size):
    raw = rng.randint(0, 256, size=(batch_size, IMAGE_HEI, IMAGE_WID, IMAGE_CHN)).astype(np.float32)
    images = ((raw - _MEAN) / _STD).astype(np.float32)
    cls = rng.randint(0, FLAGS.nb_classes, size=(batch_size,))
    labels = np.zeros((batch_size, FLAGS.nb_classes), dtype=np.float32)
    labels[np.arange(batch_size), cls] = 1.0
    return images, labels

  def build(self, enbl_trn_val_split=False, device=None):
    if not self.files:
      return super(Cifar10Dataset, self).build(enbl_trn_val_split, device)
    from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw
    images, labels = read_records(self.files)
    images, labels = torch.from_numpy(images), torch.from_numpy(labels)
    rank, size = (mgw.rank(), mgw.size()) if FLAGS.enbl_multi_gpu else (0, 1)
    seed = FLAGS.synthetic_seed + rank + (0 if self.is_train else 100003)
    device = device if device is not None else self.device

    def make(sel):
      im, lb = images[sel], labels[sel]
      if self.is_train and size > 1:             # the reference shards FILES by rank (abstract_dataset.py:80-81);
        im, lb = im[rank::size], lb[rank::size]  # records are sharded here so that every rank has data for any N
      return RecordIterator(im, lb, FLAGS.nb_classes, self.batch_size, make_transform(self.is_train),
                            shuffle=self.is_train, seed=seed, device=device)
    n = images.shape[0]
    if self.is_train and enbl_trn_val_split:
      nv = min(FLAGS.nb_smpls_val, n)
      return make(slice(nv, n)), make(slice(0, nv))   # dataset.skip(nb_smpls_val), dataset.take(nb_smpls_val)
    return make(slice(0, n))
