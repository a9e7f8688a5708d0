"""TensorFlow Saver-V2 bundle reader / writer (pocketflow_amd/utils/tf_checkpoint.py): known answers of the
primitives (crc32c, masking, varints, LevelDB block decoding with prefix-compressed keys) and round trips
through the checkpoint layer the learners use.  No TensorFlow-written file is available in this environment;
see the module docstring."""
import os
import struct

import numpy as np
import pytest

from pocketflow_amd.utils import checkpoint, tf_checkpoint as T


def test_crc32c_known_answers():
  assert T.crc32c(b'123456789') == 0xE3069283                      # the standard CRC-32C check value
  assert T.crc32c(b'') == 0
  assert T.crc32c(b'\x00' * 32) == 0x8A9136AA                      # RFC 3720 B.4 test vector
  assert T.crc32c(b'\xff' * 32) == 0x62A8AB43
  assert T.crc32c(b'6789', T.crc32c(b'12345')) == 0xE3069283       # incremental
  c = T.crc32c(b'foo')
  assert T.mask_crc(c) != c and T.mask_crc(c) == (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def test_varint_and_proto_wire_format():
  for v in (0, 1, 127, 128, 300, 2 ** 32 + 5, 2 ** 63):
    enc = T._put_varint(v)
    assert T._get_varint(enc, 0) == (v, len(enc))
  # BundleEntryProto {dtype: DT_FLOAT(1), shape {dim{size:3} dim{size:5}}, offset: 128, size: 60, crc32c: 0xdeadbeef}
  raw = bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x03, 0x12, 0x02, 0x08, 0x05, 0x20, 0x80, 0x01, 0x28, 0x3C,
               0x35]) + struct.pack('<I', 0xdeadbeef)
  e = T._parse_proto(raw)
  assert e[1] == [1] and T._parse_shape(e[2][0]) == (3, 5) and e[4] == [128] and e[5] == [60] and e[6] == [0xdeadbeef]


def test_block_decoding_with_prefix_compression():
  # keys "model/a", "model/ab", "model/b" with shared prefixes 0, 7, 6; restart array [0]
  body = b''.join([
      bytes([0, 7, 1]) + b'model/a' + b'X',
      bytes([7, 1, 2]) + b'b' + b'YZ',
      bytes([6, 1, 0]) + b'b'])
  block = body + struct.pack('<I', 0) + struct.pack('<I', 1)
  assert list(T._block_entries(block)) == [(b'model/a', b'X'), (b'model/ab', b'YZ'), (b'model/b', b'')]


def test_bundle_round_trip_and_corruption_detection(tmp_path):
  rng = np.random.RandomState(0)
  values = {'model/resnet_model/conv2d/kernel': rng.randn(3, 3, 4, 8).astype(np.float32),
            'model/resnet_model/batch_normalization/gamma': rng.rand(8).astype(np.float32),
            'model/resnet_model/dense/bias': np.zeros(10, np.float32), 'global_step': np.array(1234, np.int64),
            'model/scalar': np.array(0.5, np.float32), 'model/flags': np.array([True, False]),
            'model/half': rng.randn(7).astype(np.float16)}
  for i in range(150):                                             # several data blocks + a multi-entry index block
    values['model/extra/v%03d' % i] = rng.randn(i % 5 + 1).astype(np.float32)
  prefix = str(tmp_path / 'model.ckpt-7')
  T.write_bundle(values, prefix)
  assert os.path.exists(prefix + '.index') and os.path.exists(prefix + '.data-00000-of-00001')
  back = T.read_bundle(prefix, verify_data_crc=True)
  assert sorted(back) == sorted(values)
  for k, v in values.items():
    assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
  only = T.read_bundle(prefix, names=['global_step'])
  assert list(only) == ['global_step'] and int(only['global_step']) == 1234
  # a flipped byte in the index is caught by the block checksum, one in the data by the tensor checksum
  raw = bytearray(open(prefix + '.index', 'rb').read())
  raw[10] ^= 0x40
  open(prefix + '.index', 'wb').write(bytes(raw))
  with pytest.raises(ValueError, match='checksum'):
    T.read_index(prefix + '.index')
  raw[10] ^= 0x40
  open(prefix + '.index', 'wb').write(bytes(raw))
  data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
  data[5] ^= 0x01
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
  with pytest.raises(ValueError, match='tensor checksum'):
    T.read_bundle(prefix, verify_data_crc=True)


def test_bfloat16_tensors_are_widened(tmp_path):
  x = np.array([1.0, -2.5, 3.140625], np.float32)
  prefix = str(tmp_path / 'b')
  T.write_bundle({'w': x}, prefix)
  # rewrite the entry as DT_BFLOAT16 over the upper halves of the float32 words
  bf = (x.view(np.uint32) >> 16).astype('<u2')
  open(prefix + '.data-00000-of-00001', 'wb').write(bf.tobytes())
  entry = (T._emit(1, 0, T._put_varint(T.DT_BFLOAT16)) + T._emit(2, 2, bytes([4, 0x12, 0x02, 0x08, 0x03]))
           + T._emit(5, 0, T._put_varint(6)))
  header = T._emit(1, 0, T._put_varint(1))
  blk = T._build_block([(b'', header), (b'w', entry)])
  out = bytearray(blk + b'\x00' + struct.pack('<I', T.mask_crc(T.crc32c(blk + b'\x00'))))
  h0 = T._put_varint(0) + T._put_varint(len(blk))
  meta = T._build_block([])
  moff = len(out)
  out += meta + b'\x00' + struct.pack('<I', T.mask_crc(T.crc32c(meta + b'\x00')))
  idx = T._build_block([(b'w', h0)])
  ioff = len(out)
  out += idx + b'\x00' + struct.pack('<I', T.mask_crc(T.crc32c(idx + b'\x00')))
  footer = T._put_varint(moff) + T._put_varint(len(meta)) + T._put_varint(ioff) + T._put_varint(len(idx))
  out += footer + b'\x00' * (T.FOOTER_LEN - 8 - len(footer)) + struct.pack('<Q', T.TABLE_MAGIC)
  open(prefix + '.index', 'wb').write(bytes(out))
  back = T.read_bundle(prefix)
  assert back['w'].dtype == np.float32 and np.array_equal(back['w'], x)        # these values are exact in bf16


def test_checkpoint_layer_reads_and_writes_tf_bundles(tmp_path):
  vals = {'model/a/kernel': np.arange(24, dtype=np.float32).reshape(1, 1, 4, 6), 'model/b': np.ones(3, np.float32)}
  p = checkpoint.save(vals, str(tmp_path / 'models' / 'model.ckpt'), 42, fmt='tf')
  assert p.endswith('model.ckpt-42') and not os.path.exists(p + '.npz')
  assert checkpoint.latest_checkpoint(str(tmp_path / 'models')) == p        # via the TF `checkpoint` state file
  back = checkpoint.load(p)
  assert sorted(back) == sorted(vals) and all(np.array_equal(back[k], vals[k]) for k in vals)
  with pytest.raises(ValueError):
    checkpoint.save(vals, str(tmp_path / 'x' / 'm'), fmt='hdf5')


def test_varstore_restores_from_a_tf_bundle_with_extra_slots(tmp_path):
  """A pre-trained archive also holds optimiser slots and global_step: restoring ignores them by name."""
  import torch
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd.graph import Graph
  from pocketflow_amd.nets.lenet_at_cifar10 import _LeNet
  g = Graph('model', 'cpu', torch.float32)
  _LeNet(g, (32, 32, 3), 10)
  g.finalize(seed=3, requires_grad=False)
  vals = g.store.export_numpy()
  extra = dict(vals)
  extra['global_step'] = np.array(99, np.int64)
  extra['model/conv1/kernel/Momentum'] = np.zeros_like(vals['model/conv1/kernel'])
  p = checkpoint.save(extra, str(tmp_path / 'models' / 'model.ckpt'), 99, fmt='tf')
  g2 = Graph('model', 'cpu', torch.float32)
  _LeNet(g2, (32, 32, 3), 10)
  g2.finalize(seed=4, requires_grad=False)
  g2.store.load_numpy(checkpoint.load(p), strict=True)
  for k, v in g2.store.export_numpy().items():
    assert np.array_equal(v, vals[k]), k
