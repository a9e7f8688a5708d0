"""Streaming iterator over ImageNet-style TFRecord shards (stands in for the tf.data chain of the reference,
datasets/abstract_dataset.py:76-111):

  list_files(shuffle) [-> shard(size, rank)] -> parallel_interleave(TFRecordDataset, cycle_length)
  -> map(parse_fn, nb_threads) [-> take / skip(nb_smpls_val)] -> shuffle_and_repeat(buffer_size) -> batch -> prefetch

Host side: a producer thread walks the shards (round-robin over `cycle_length` open files), keeps a seeded shuffle
buffer of serialized records and hands batches of them to a pool of `nb_threads` decode workers (Pillow releases the
GIL inside libjpeg); up to `prefetch_size` decoded batches wait in a queue.  Device side: `get_next()` uploads the
packed uint8 crops of one batch and runs ONE kernel (pf_image_resize_bilinear) that produces the float NHWC batch.
Everything random (file order, shuffle buffer, crop windows, flips) derives from `seed`, unlike the reference.
"""
from __future__ import annotations

import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional

import numpy as np
import torch

from pocketflow_amd.datasets import imagenet_preprocessing as prep
from pocketflow_amd.datasets.tfrecord import parse_example, read_records


def parse_example_proto(example_serialized: bytes):
  """(jpeg bytes, label int, bbox [n, 4] rows ymin, xmin, ymax, xmax) -- reference ilsvrc12_dataset.py:39-73."""
  f = parse_example(example_serialized)
  label = int(f['image/class/label'][0]) if len(f.get('image/class/label', [])) else -1
  cols = [np.asarray(f.get('image/object/bbox/' + k, np.zeros(0, np.float32)), dtype=np.float32)
          for k in ('ymin', 'xmin', 'ymax', 'xmax')]
  n = min(len(c) for c in cols)
  bbox = np.stack([c[:n] for c in cols], axis=1) if n else np.zeros((0, 4), np.float32)
  enc = f.get('image/encoded', [b''])
  return (enc[0] if enc else b''), label, bbox


class TFRecordImageIterator(object):  # pylint: disable=too-many-instance-attributes
  def __init__(self, files: List[str], batch_size: int, is_train: bool, nb_classes: int, seed: int, device=None,
               output_hw=(224, 224), skip: int = 0, take: Optional[int] = None, cycle_length: int = 4,
               nb_threads: int = 8, buffer_size: int = 1024, prefetch_size: int = 8, dtype=torch.float32):
    if not files:
      raise ValueError('no TFRecord files')
    self.files, self.batch_size, self.is_train, self.nb_classes, self.seed = list(files), batch_size, is_train, nb_classes, seed
    self.output_hw, self.skip, self.take = output_hw, int(skip), take
    self.cycle_length, self.nb_threads = max(1, cycle_length), max(1, nb_threads)
    self.buffer_size, self.prefetch_size, self.dtype = max(1, buffer_size), max(1, prefetch_size), dtype
    self.device = torch.device(device) if device is not None else torch.device('cpu')
    self._thread = None
    self.reset()

  # -- the reference's iterator surface ------------------------------------------------------------------------
  def to(self, device):
    self.device = torch.device(device)
    return self

  def reset(self):
    self.__stop()
    self._queue = queue.Queue(maxsize=self.prefetch_size)
    self._stop = threading.Event()
    self._thread = threading.Thread(target=self.__produce, args=(self._queue, self._stop), daemon=True)
    self._thread.start()

  def get_next(self):
    item = self._queue.get()
    if isinstance(item, BaseException):
      raise item
    crops, descs, labels = item
    images = prep.preprocess_batch(crops, descs, self.output_hw[0], self.output_hw[1], self.device, dtype=self.dtype)
    onehot = torch.zeros((len(labels), self.nb_classes), dtype=torch.float32)
    idx = torch.as_tensor(labels, dtype=torch.int64)
    valid = (idx >= 0) & (idx < self.nb_classes)                 # tf.one_hot: out-of-range indices give all-zero rows
    onehot[torch.arange(len(labels))[valid], idx[valid]] = 1.0
    return images, onehot.to(self.device, non_blocking=True)

  def close(self):
    self.__stop()

  def __del__(self):
    try:
      self.__stop()
    except Exception:   # pylint: disable=broad-except
      pass

  # -- producer ----------------------------------------------------------------------------------------------------
  def __stop(self):
    if self._thread is not None:
      self._stop.set()
      try:
        while True:
          self._queue.get_nowait()
      except queue.Empty:
        pass
      self._thread.join(timeout=5.0)
      self._thread = None

  def __records_one_pass(self, rng):
    """One pass over the shards: seeded file order, round-robin interleave of cycle_length open files, take/skip."""
    order = list(self.files)
    if self.is_train:
      rng.shuffle(order)
    pending = [read_records(p) for p in order]
    active, n = [], 0
    while pending or active:
      while pending and len(active) < self.cycle_length:
        active.append(pending.pop(0))
      for it in list(active):
        try:
          rec = next(it)
        except StopIteration:
          active.remove(it)
          continue
        n += 1
        if n <= self.skip:
          continue
        if self.take is not None and n - self.skip > self.take:
          return
        yield rec

  def __shuffled_forever(self, rng):
    """shuffle_and_repeat(buffer_size): a reservoir of serialized records, drained at the end of every pass."""
    buf = []
    while True:
      got = False
      for rec in self.__records_one_pass(rng):
        got = True
        if not self.is_train:
          yield rec
          continue
        buf.append(rec)
        if len(buf) >= self.buffer_size:
          k = rng.randint(0, len(buf))
          buf[k], buf[-1] = buf[-1], buf[k]
          yield buf.pop()
      if not got:
        raise ValueError('the TFRecord files hold no records (after skip / take)')
      if not self.is_train:
        continue
      while buf:                                                   # drain at the end of a pass (epochs do not mix)
        k = rng.randint(0, len(buf))
        buf[k], buf[-1] = buf[-1], buf[k]
        yield buf.pop()

  def __decode(self, args):
    rec, sub_seed = args
    jpeg, label, bbox = parse_example_proto(rec)
    img, desc = prep.preprocess_image(jpeg, bbox, self.output_hw[0], self.output_hw[1], 3, self.is_train,
                                      np.random.RandomState(sub_seed))
    return np.ascontiguousarray(img), desc, label

  def __produce(self, out_queue, stop):
    try:
      rng = np.random.RandomState(self.seed)
      stream = self.__shuffled_forever(rng)
      counter = 0
      with ThreadPoolExecutor(max_workers=self.nb_threads) as pool:
        while not stop.is_set():
          jobs = []
          for _ in range(self.batch_size):
            jobs.append((next(stream), (self.seed * 1000003 + counter) % (2 ** 31 - 1)))
            counter += 1
          done = list(pool.map(self.__decode, jobs))
          item = ([d[0] for d in done], [d[1] for d in done], [d[2] for d in done])
          while not stop.is_set():
            try:
              out_queue.put(item, timeout=0.1)
              break
            except queue.Full:
              continue
    except BaseException as err:   # pylint: disable=broad-except
      if not stop.is_set():
        out_queue.put(err)
