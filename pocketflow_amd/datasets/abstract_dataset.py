"""Seeded synthetic stand-in for the reference's tf.data pipelines (datasets/abstract_dataset.py).

Only the OUTPUT CONTRACT of the reference pipelines is on the hot path: an
iterator whose `get_next()` yields `(images float32 NHWC, labels one-hot float32)`.  The benchmark
configurations use synthetic batches, so a fixed pool of seeded batches is generated once, kept
resident in HBM and cycled; rank r draws from seed 1234 + r (the reference shards files by rank,
abstract_dataset.py:80-81).  Flags of the real pipelines are kept so command lines stay valid.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np
import torch

from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.utils import misc_utils  # noqa: F401  (defines --enbl_multi_gpu, read by build())

flags.DEFINE_string('data_disk', 'local', 'data disk\'s location (\'local\' / \'hdfs\')')
flags.DEFINE_string('data_hdfs_host', None, 'HDFS host for data files')
flags.DEFINE_string('data_dir_local', None, 'data directory - local')
flags.DEFINE_string('data_dir_hdfs', None, 'data directory - HDFS')
flags.DEFINE_integer('cycle_length', 4, '# of datasets to interleave from in parallel')
flags.DEFINE_integer('nb_threads', 8, '# of threads for preprocessing the dataset')
flags.DEFINE_integer('buffer_size', 1024, '# of elements to be buffered when prefetching')
flags.DEFINE_integer('prefetch_size', 8, '# of mini-batches to be buffered when prefetching')
flags.DEFINE_integer('synthetic_pool', 8, '# of distinct synthetic mini-batches that are cycled')
flags.DEFINE_integer('synthetic_seed', 1234, 'base seed of the synthetic data (rank is added)')


class SyntheticIterator(object):
  """`iterator.get_next()` of the reference: (images NHWC float32, one-hot labels float32)."""

  def __init__(self, make_batch, batch_size, pool, seed, device):
    self.batches = []
    rng = np.random.RandomState(seed)
    for _ in range(pool):
      images, labels = make_batch(rng, batch_size)
      self.batches.append((torch.from_numpy(images), torch.from_numpy(labels)))
    self.device = None
    self.idx = 0
    if device is not None:
      self.to(device)

  def to(self, device):
    self.device = torch.device(device)
    self.batches = [(i.to(self.device), l.to(self.device)) for i, l in self.batches]
    return self

  def get_next(self):
    b = self.batches[self.idx % len(self.batches)]
    self.idx += 1
    return b

  def reset(self):
    self.idx = 0


class AbstractDataset(ABC):
  """Same surface as the reference's AbstractDataset: `build(enbl_trn_val_split)` -> iterator."""

  def __init__(self, is_train):
    self.is_train = is_train
    self.batch_size = None
    self.device = None

  @abstractmethod
  def make_batch(self, rng, batch_size):
    """Return (images [B,H,W,C] float32, labels one-hot [B,nb_classes] float32)."""

  def build(self, enbl_trn_val_split=False, device=None):
    from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw
    rank = mgw.rank() if FLAGS.enbl_multi_gpu else 0
    seed = FLAGS.synthetic_seed + rank + (0 if self.is_train else 100003)
    device = device if device is not None else self.device
    it = SyntheticIterator(self.make_batch, self.batch_size, FLAGS.synthetic_pool, seed, device)
    if enbl_trn_val_split:
      it_val = SyntheticIterator(self.make_batch, self.batch_size, FLAGS.synthetic_pool, seed + 50021, device)
      return it, it_val
    return it
