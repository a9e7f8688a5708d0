"""TFRecord files and tf.train.Example messages without TensorFlow (the on-disk format of the reference's ILSVRC-12
pipeline: `tf.data.TFRecordDataset` + `tf.parse_single_example`, datasets/ilsvrc12_dataset.py:39-73, 126).

Record framing (tensorflow/core/lib/io/record_writer.cc):
    uint64 length | uint32 masked_crc32c(length) | byte data[length] | uint32 masked_crc32c(data)
tf.train.Example (tensorflow/core/example/{example,feature}.proto):
    Example { Features features = 1 }      Features { map<string, Feature> feature = 1 }
    Feature { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3 } }
    BytesList { repeated bytes value = 1 }   FloatList { repeated float value = 1 [packed] }
    Int64List { repeated int64 value = 1 [packed] }
Both packed and unpacked encodings of the repeated scalars are accepted on input; output is packed, like TF's.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Union

import numpy as np

from pocketflow_amd.utils.tf_checkpoint import _get_varint, _put_varint, crc32c, mask_crc

FeatureValue = Union[List[bytes], np.ndarray]


# -- framing --------------------------------------------------------------------------------------------------------------
def read_records(path: str, verify_crc: bool = False) -> Iterator[bytes]:
  """Yield the payload of every record of one TFRecord file.  `verify_crc` checks both checksums (pure-Python
  crc32c: ~30 ms per 100 kB record, so it is off by default; the length checksum is always checked)."""
  with open(path, 'rb') as f:
    while True:
      head = f.read(12)
      if not head:
        return
      if len(head) != 12:
        raise ValueError('%s: truncated record header' % path)
      length, len_crc = struct.unpack('<QI', head)
      if mask_crc(crc32c(head[:8])) != len_crc:
        raise ValueError('%s: corrupted record length' % path)
      data = f.read(length)
      tail = f.read(4)
      if len(data) != length or len(tail) != 4:
        raise ValueError('%s: truncated record' % path)
      if verify_crc and mask_crc(crc32c(data)) != struct.unpack('<I', tail)[0]:
        raise ValueError('%s: corrupted record data' % path)
      yield data


def write_records(path: str, records) -> int:
  n = 0
  with open(path, 'wb') as f:
    for data in records:
      head = struct.pack('<Q', len(data))
      f.write(head + struct.pack('<I', mask_crc(crc32c(head))) + data + struct.pack('<I', mask_crc(crc32c(data))))
      n += 1
  return n


# -- tf.train.Example ---------------------------------------------------------------------------------------------------------
def _fields(buf: bytes):
  pos, n = 0, len(buf)
  while pos < n:
    key, pos = _get_varint(buf, pos)
    field, wire = key >> 3, key & 7
    if wire == 0:
      val, pos = _get_varint(buf, pos)
    elif wire == 1:
      val, pos = buf[pos:pos + 8], pos + 8
    elif wire == 2:
      ln, pos = _get_varint(buf, pos)
      val, pos = buf[pos:pos + ln], pos + ln
    elif wire == 5:
      val, pos = buf[pos:pos + 4], pos + 4
    else:
      raise ValueError('unsupported protobuf wire type %d' % wire)
    yield field, wire, val


def _parse_feature(buf: bytes) -> FeatureValue:
  for field, wire, val in _fields(buf):
    if field == 1:                                      # BytesList
      return [bytes(v) for f, w, v in _fields(val) if f == 1]
    if field == 2:                                      # FloatList
      out = []
      for f, w, v in _fields(val):
        if f == 1:
          out.append(np.frombuffer(v, dtype='<f4'))     # packed run (wire 2) or a single fixed32 (wire 5)
      return np.concatenate(out).astype(np.float32) if out else np.zeros(0, np.float32)
    if field == 3:                                      # Int64List
      out = []
      for f, w, v in _fields(val):
        if f != 1:
          continue
        if w == 0:
          out.append(v)
        else:
          pos = 0
          while pos < len(v):
            x, pos = _get_varint(v, pos)
            out.append(x)
      return np.array([x - (1 << 64) if x >= (1 << 63) else x for x in out], dtype=np.int64)
  return []                                             # Feature with no list set


def parse_example(buf: bytes) -> Dict[str, FeatureValue]:
  """tf.train.Example bytes -> {feature name: list of bytes | float32 array | int64 array}."""
  features = {}
  for field, wire, val in _fields(buf):
    if field != 1:
      continue
    for f2, w2, entry in _fields(val):                  # map<string, Feature> entries
      if f2 != 1:
        continue
      key, feat = None, b''
      for f3, w3, v3 in _fields(entry):
        if f3 == 1:
          key = bytes(v3).decode('utf-8')
        elif f3 == 2:
          feat = v3
      features[key] = _parse_feature(feat)
  return features


def _ld(field: int, payload: bytes) -> bytes:
  return _put_varint((field << 3) | 2) + _put_varint(len(payload)) + payload


def make_example(features: Dict[str, FeatureValue]) -> bytes:
  """{name: bytes | list of bytes | float array | int array} -> serialized tf.train.Example (sorted keys, as TF)."""
  entries = b''
  for key in sorted(features):
    v = features[key]
    if isinstance(v, (bytes, bytearray)):
      v = [bytes(v)]
    if isinstance(v, list) and (not v or isinstance(v[0], (bytes, bytearray))):
      feat = _ld(1, b''.join(_ld(1, bytes(x)) for x in v))
    else:
      arr = np.asarray(v)
      if arr.dtype.kind == 'f':
        feat = _ld(2, _ld(1, arr.astype('<f4').tobytes()) if arr.size else b'')
      else:
        packed = b''.join(_put_varint(int(x) & ((1 << 64) - 1)) for x in arr.reshape(-1))
        feat = _ld(3, _ld(1, packed) if arr.size else b'')
    entries += _ld(1, _ld(1, key.encode('utf-8')) + _ld(2, feat))
  return _ld(1, entries)
