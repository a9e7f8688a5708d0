"""In-memory record iterator for real datasets small enough to live in HBM (CIFAR-10: 150 MB as uint8).

Stands in for the reference's tf.data chain `list_files -> interleave -> map(parse_fn) -> shuffle_and_repeat
-> batch -> prefetch -> one_shot_iterator` (datasets/abstract_dataset.py:76-111): `get_next()` yields
(images float32 NHWC, labels one-hot float32) for ever.  The whole subset is kept as uint8 on the device;
shuffling is a seeded full permutation per epoch (the reference shuffles inside a 1024-element buffer and never
seeds: SURVEY section 4), decoding / augmentation / standardisation run batch-wise on the device.
"""
from __future__ import annotations

from typing import Callable

import torch


class RecordIterator(object):
  def __init__(self, images_u8: torch.Tensor, labels: torch.Tensor, nb_classes: int, batch_size: int,
               transform: Callable[[torch.Tensor, torch.Generator], torch.Tensor], shuffle: bool, seed: int,
               device=None):
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and labels.dim() == 1
    self.images, self.labels = images_u8, labels.long()
    self.nb_classes, self.batch_size, self.transform, self.shuffle, self.seed = nb_classes, batch_size, transform, shuffle, seed
    self.device = torch.device('cpu')
    self.reset()
    if device is not None:
      self.to(device)

  def __len__(self):
    return self.images.shape[0]

  def to(self, device):
    self.device = torch.device(device)
    self.images, self.labels = self.images.to(self.device), self.labels.to(self.device)
    self.reset()
    return self

  def reset(self):
    self.gen = torch.Generator(device='cpu')
    self.gen.manual_seed(self.seed)
    self.epoch, self.pos = 0, 0
    self._order = self._new_order()

  def _new_order(self) -> torch.Tensor:
    n = len(self)
    order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
    return order.to(self.device)

  def get_next(self):
    n, b = len(self), self.batch_size
    idx = []
    need = b
    while need > 0:                                     # repeat(): batches run across epoch boundaries
      take = min(need, n - self.pos)
      idx.append(self._order[self.pos:self.pos + take])
      self.pos += take
      need -= take
      if self.pos == n:
        self.epoch, self.pos = self.epoch + 1, 0
        self._order = self._new_order()
    idx = torch.cat(idx) if len(idx) > 1 else idx[0]
    images = self.transform(self.images[idx], self.gen)
    labels = torch.nn.functional.one_hot(self.labels[idx], self.nb_classes).to(torch.float32)
    return images, labels
