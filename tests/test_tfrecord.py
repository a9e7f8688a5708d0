"""TFRecord framing and tf.train.Example codec (pocketflow_amd/datasets/tfrecord.py) against the protobuf runtime
(message classes built from example.proto / feature.proto definitions restated as descriptors) and hand-assembled
bytes.  No GPU, no TensorFlow."""
import struct

import numpy as np
import pytest

from pocketflow_amd.datasets import tfrecord as T


def _example_classes():
  """tf.train.Example & friends as dynamic protobuf messages (tensorflow/core/example/{example,feature}.proto)."""
  from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
  fd = descriptor_pb2.FileDescriptorProto(name='pf_example_test.proto', package='pftest', syntax='proto3')
  F = descriptor_pb2.FieldDescriptorProto

  def msg(name, fields):
    m = fd.message_type.add(name=name)
    for (fname, num, ftype, label, tname) in fields:
      f = m.field.add(name=fname, number=num, type=ftype, label=label)
      if tname:
        f.type_name = tname
    return m
  msg('BytesList', [('value', 1, F.TYPE_BYTES, F.LABEL_REPEATED, None)])
  msg('FloatList', [('value', 1, F.TYPE_FLOAT, F.LABEL_REPEATED, None)])
  msg('Int64List', [('value', 1, F.TYPE_INT64, F.LABEL_REPEATED, None)])
  feat = msg('Feature', [('bytes_list', 1, F.TYPE_MESSAGE, F.LABEL_OPTIONAL, '.pftest.BytesList'),
                         ('float_list', 2, F.TYPE_MESSAGE, F.LABEL_OPTIONAL, '.pftest.FloatList'),
                         ('int64_list', 3, F.TYPE_MESSAGE, F.LABEL_OPTIONAL, '.pftest.Int64List')])
  feat.oneof_decl.add(name='kind')
  for f in feat.field:
    f.oneof_index = 0
  feats = msg('Features', [('feature', 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, '.pftest.Features.FeatureEntry')])
  entry = feats.nested_type.add(name='FeatureEntry')
  entry.field.add(name='key', number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
  entry.field.add(name='value', number=2, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name='.pftest.Feature')
  entry.options.map_entry = True
  msg('Example', [('features', 1, F.TYPE_MESSAGE, F.LABEL_OPTIONAL, '.pftest.Features')])
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  get = getattr(message_factory, 'GetMessageClass', None)
  if get is None:
    factory = message_factory.MessageFactory(pool)
    get = factory.GetPrototype
  return get(pool.FindMessageTypeByName('pftest.Example'))


FEATURES = {
    'image/encoded': [b'\xff\xd8\xff\xe0 not really a jpeg \x00\x01' * 7],
    'image/class/label': np.array([417], dtype=np.int64),
    'image/class/text': [b'balloon'],
    'image/object/bbox/xmin': np.array([0.1, 0.25], dtype=np.float32),
    'image/object/bbox/ymin': np.array([0.0, 0.5], dtype=np.float32),
    'image/object/bbox/xmax': np.array([0.9, 0.75], dtype=np.float32),
    'image/object/bbox/ymax': np.array([1.0, 0.875], dtype=np.float32),
    'negative': np.array([-1, -(1 << 40), 1 << 62], dtype=np.int64),
    'empty': np.zeros(0, np.float32),
}


def _same(a, b):
  for k in b:
    if isinstance(b[k], list):
      assert a[k] == b[k], k
    else:
      assert np.array_equal(a[k], b[k]) and a[k].dtype == b[k].dtype, k


def test_example_written_by_the_protobuf_runtime_is_parsed():
  Example = _example_classes()
  ex = Example()
  for k, v in FEATURES.items():
    f = ex.features.feature[k]
    if isinstance(v, list):
      f.bytes_list.value.extend(v)
    elif v.dtype.kind == 'f':
      f.float_list.value.extend(v.tolist())
      if not v.size:
        f.float_list.SetInParent()
    else:
      f.int64_list.value.extend(v.tolist())
  got = T.parse_example(ex.SerializeToString())
  assert set(got) == set(FEATURES)
  _same(got, FEATURES)


def test_example_written_here_is_parsed_by_the_protobuf_runtime():
  Example = _example_classes()
  ex = Example()
  ex.ParseFromString(T.make_example(FEATURES))
  assert set(ex.features.feature) == set(FEATURES)
  for k, v in FEATURES.items():
    f = ex.features.feature[k]
    if isinstance(v, list):
      assert list(f.bytes_list.value) == v
    elif v.dtype.kind == 'f':
      assert np.array_equal(np.array(f.float_list.value, np.float32), v)
    else:
      assert list(f.int64_list.value) == v.tolist()
  _same(T.parse_example(T.make_example(FEATURES)), FEATURES)


def test_unpacked_repeated_scalars_are_accepted():
  # Int64List with two unpacked varints (field 1, wire 0) and FloatList with one unpacked fixed32 (wire 5)
  ints = b'\x08\x05\x08\x07'
  floats = b'\x0d' + struct.pack('<f', 2.5)
  def entry(key, feat):
    e = T._ld(1, key) + T._ld(2, feat)
    return T._ld(1, e)
  buf = T._ld(1, entry(b'i', T._ld(3, ints)) + entry(b'f', T._ld(2, floats)))
  got = T.parse_example(buf)
  assert got['i'].tolist() == [5, 7] and got['f'].tolist() == [2.5]


def test_tfrecord_framing_round_trip_and_corruption(tmp_path):
  recs = [b'', b'a', bytes(range(256)) * 33, T.make_example(FEATURES)]
  path = str(tmp_path / 'train-00000-of-00001')
  assert T.write_records(path, recs) == 4
  assert list(T.read_records(path, verify_crc=True)) == recs
  raw = bytearray(open(path, 'rb').read())
  # known answer for the first (empty) record: length 0, masked crc32c of eight zero bytes, masked crc32c of b''
  assert bytes(raw[:8]) == b'\x00' * 8 and struct.unpack('<I', raw[12:16])[0] == T.mask_crc(0)
  raw[60] ^= 0xFF                                           # inside the payload of the third record (starts at 45)
  open(path, 'wb').write(bytes(raw))
  assert len(list(T.read_records(path))) == 4               # payload checksums are not verified by default
  with pytest.raises(ValueError):
    list(T.read_records(path, verify_crc=True))
  raw[16] ^= 0x01                                           # length field of the second record
  open(path, 'wb').write(bytes(raw))
  with pytest.raises(ValueError):
    list(T.read_records(path))
