"""TensorFlow Saver-V2 ("tensor bundle") checkpoints: reader and writer in pure Python / NumPy.

SURVEY section 8(f) row 1 -- the on-disk format the path starts from: the reference downloads
`models_<model>_at_<dataset>.tar.gz` and restores `./models/model.ckpt-<step>.{index,data-00000-of-00001}`
with tf.train.Saver (learners/abstract_learner.py:90,105-125; docs/docs/pre_trained_models.md:7-23), and the
teacher is a renamed copy of it (learners/distillation_helper.py:122-145).  TensorFlow is not available, so
the format is read directly:

  <prefix>.index                an SSTable (TensorFlow's copy of the LevelDB table format,
                                tensorflow/core/lib/io/table*.cc, uncompressed blocks): key "" -> BundleHeaderProto,
                                key <variable name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
  <prefix>.data-0000k-of-0000n  raw little-endian tensor bytes, row-major, at [offset, offset + size)

Variable names and layouts in such a checkpoint are exactly the ones the VarStore speaks
(`model/resnet_model/conv2d_3/kernel`, HWIO), so `VarStore.load_numpy(read_bundle(prefix))` restores a
pre-trained PocketFlow model; optimiser slots and `global_step` in the file are ignored by name.

The writer produces bundles TensorFlow's BundleReader accepts (header, masked crc32c of every block and tensor),
so quantised / pruned weights can be handed to the reference's `tools/conversion/*` scripts.

Pinning note: no TensorFlow-written checkpoint exists in this environment (no network, no TF), so the reader is
tested against bundles produced by this writer and against hand-assembled byte strings of the documented block /
protobuf encodings (tests/test_tf_checkpoint.py) -- not yet against a file written by TensorFlow itself.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
BLOCK_TRAILER_LEN = 5            # 1-byte compression type + 4-byte masked crc32c

# tensorflow/core/framework/types.proto
DT = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
      17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
DT_BFLOAT16 = 14
DT_OF = {np.dtype(v): k for k, v in DT.items()}


# -- crc32c (Castagnoli), masked as in tensorflow/core/lib/hash/crc32c.h ----------------------------------

def _make_table():
  poly = 0x82F63B78
  tbl = []
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ poly if c & 1 else c >> 1
    tbl.append(c)
  return tbl


_CRC_TABLE = _make_table()


def _crc32c_bytes(data, crc: int = 0) -> int:
  """Byte-at-a-time table CRC (short inputs: record headers, index blocks)."""
  c = crc ^ 0xFFFFFFFF
  tbl = _CRC_TABLE
  for b in data:
    c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
  return c ^ 0xFFFFFFFF


# -- long inputs (tensor payloads: ~100 MB per ResNet-50 checkpoint).  A CRC is sequential in its input, but linear
# over GF(2): the buffer is cut into `_LANES` equal chunks whose CRCs advance in lockstep as one NumPy vector (one
# table gather per byte POSITION, not per byte), and the chunk CRCs are chained with the "append n zero bytes"
# operator (a 32x32 bit matrix, built by repeated squaring like zlib's crc32_combine).

_LANES = 8192
_CRC_TABLE_NP = np.array(_CRC_TABLE, dtype=np.uint32)


def _gf2_times(mat, vec: int) -> int:
  out, i = 0, 0
  while vec:
    if vec & 1:
      out ^= mat[i]
    vec >>= 1
    i += 1
  return out


def _gf2_square(mat):
  return [_gf2_times(mat, mat[i]) for i in range(32)]


def _zeros_operator(nbytes: int):
  """Matrix that maps crc(A) to the raw CRC state after `nbytes` further zero bytes."""
  odd = [0x82F63B78] + [1 << i for i in range(31)]         # one zero BIT
  even = _gf2_square(odd)                                   # two bits
  odd = _gf2_square(even)                                   # four bits
  result = None
  n = nbytes
  while n:
    even = _gf2_square(odd)                                 # first pass: one zero byte
    if n & 1:
      result = even if result is None else [_gf2_times(even, result[i]) for i in range(32)]
    n >>= 1
    if not n:
      break
    odd = _gf2_square(even)
    if n & 1:
      result = odd if result is None else [_gf2_times(odd, result[i]) for i in range(32)]
    n >>= 1
  return result


def _crc32c_combine(crc1: int, crc2: int, op) -> int:
  """crc(A || B) from crc(A), crc(B) and the zeros operator of len(B) (zlib crc32_combine)."""
  return _gf2_times(op, crc1) ^ crc2


def crc32c(data, crc: int = 0) -> int:
  buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
  n = buf.size
  if n < (1 << 16):
    return _crc32c_bytes(buf.tolist(), crc)
  chunk = n // _LANES
  body = buf[:chunk * _LANES].reshape(_LANES, chunk)
  c = np.full(_LANES, 0xFFFFFFFF, dtype=np.uint32)
  tbl = _CRC_TABLE_NP
  for j in range(chunk):
    c = tbl[(c ^ body[:, j]) & 0xFF] ^ (c >> 8)
  c ^= 0xFFFFFFFF                                          # per-chunk CRCs (each from a zero start value)
  op = _zeros_operator(chunk)
  total = crc
  for v in c.tolist():
    total = _crc32c_combine(total, int(v), op)
  tail = buf[chunk * _LANES:]
  if tail.size:
    total = _crc32c_bytes(tail.tolist(), total)
  return total


def mask_crc(crc: int) -> int:
  return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


# -- varints / protobuf wire format ------------------------------------------------------------------------------

def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
  result, shift = 0, 0
  while True:
    b = buf[pos]
    pos += 1
    result |= (b & 0x7F) << shift
    if not b & 0x80:
      return result, pos
    shift += 7


def _put_varint(v: int) -> bytes:
  out = bytearray()
  while True:
    b = v & 0x7F
    v >>= 7
    if v:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _parse_proto(buf: bytes) -> Dict[int, List]:
  """field number -> list of raw values (int for varint / fixed, bytes for length-delimited)."""
  out: Dict[int, List] = {}
  pos = 0
  while pos < len(buf):
    key, pos = _get_varint(buf, pos)
    field, wire = key >> 3, key & 7
    if wire == 0:
      val, pos = _get_varint(buf, pos)
    elif wire == 1:
      val = struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    elif wire == 2:
      n, pos = _get_varint(buf, pos)
      val = bytes(buf[pos:pos + n])
      pos += n
    elif wire == 5:
      val = struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    else:
      raise ValueError('unsupported protobuf wire type %d' % wire)
    out.setdefault(field, []).append(val)
  return out


def _zigzag_free_int64(v: int) -> int:
  """protobuf int64 varints are two's complement in 64 bits."""
  return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
  dims = []
  for d in _parse_proto(buf).get(2, []):               # TensorShapeProto.dim
    f = _parse_proto(d)
    dims.append(_zigzag_free_int64(f.get(1, [0])[0]))
  return tuple(dims)


# -- SSTable ---------------------------------------------------------------------------------------------------------

def _read_block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
  contents = buf[offset:offset + size]
  trailer = buf[offset + size:offset + size + BLOCK_TRAILER_LEN]
  if len(contents) != size or len(trailer) != BLOCK_TRAILER_LEN:
    raise ValueError('truncated table block')
  if trailer[0] != 0:
    raise ValueError('compressed table blocks are not supported (type %d)' % trailer[0])
  if verify:
    want = struct.unpack('<I', trailer[1:5])[0]
    got = mask_crc(crc32c(contents + trailer[0:1]))
    if want != got:
      raise ValueError('table block checksum mismatch')
  return contents


def _block_entries(block: bytes):
  """Yield (key, value) of one block (prefix-compressed keys, restart array at the end)."""
  if len(block) < 4:
    raise ValueError('bad table block')
  num_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
  limit = len(block) - 4 - 4 * num_restarts
  pos, key = 0, b''
  while pos < limit:
    shared, pos = _get_varint(block, pos)
    unshared, pos = _get_varint(block, pos)
    vlen, pos = _get_varint(block, pos)
    key = key[:shared] + block[pos:pos + unshared]
    pos += unshared
    yield key, block[pos:pos + vlen]
    pos += vlen


def read_index(path: str, verify: bool = True) -> Dict[str, bytes]:
  """All (key, value) pairs of an SSTable file."""
  with open(path, 'rb') as f:
    buf = f.read()
  if len(buf) < FOOTER_LEN:
    raise ValueError('%s: too short for a table footer' % path)
  footer = buf[-FOOTER_LEN:]
  if struct.unpack('<Q', footer[-8:])[0] != TABLE_MAGIC:
    raise ValueError('%s: not a TensorFlow checkpoint index (bad magic)' % path)
  pos = 0
  _, pos = _get_varint(footer, pos)                    # metaindex handle (unused)
  _, pos = _get_varint(footer, pos)
  idx_off, pos = _get_varint(footer, pos)
  idx_size, pos = _get_varint(footer, pos)
  out: Dict[str, bytes] = {}
  for _, handle in _block_entries(_read_block(buf, idx_off, idx_size, verify)):
    off, p = _get_varint(handle, 0)
    size, p = _get_varint(handle, p)
    for key, value in _block_entries(_read_block(buf, off, size, verify)):
      out[key.decode('utf-8')] = bytes(value)
  return out


# -- bundle ------------------------------------------------------------------------------------------------------------

def _data_path(prefix: str, shard: int, num_shards: int) -> str:
  return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def is_bundle(prefix: str) -> bool:
  return os.path.exists(prefix + '.index')


def read_bundle(prefix: str, verify_data_crc: bool = False, names: Optional[List[str]] = None
                ) -> Dict[str, np.ndarray]:
  """Every tensor of `<prefix>.index` / `.data-*` as NumPy arrays keyed by variable name.
  bfloat16 tensors are widened to float32.  `names`: restrict to these variables."""
  table = read_index(prefix + '.index')
  if '' not in table:
    raise ValueError('%s.index has no bundle header' % prefix)
  header = _parse_proto(table[''])
  num_shards = header.get(1, [1])[0]
  if header.get(2, [0])[0] != 0:
    raise ValueError('big-endian bundles are not supported')
  shards: Dict[int, np.memmap] = {}
  out: Dict[str, np.ndarray] = {}
  for name, raw in table.items():
    if name == '' or (names is not None and name not in names):
      continue
    e = _parse_proto(raw)
    if 7 in e:
      raise ValueError('%s: sliced (partitioned) variables are not supported' % name)
    dtype = e.get(1, [0])[0]
    shape = _parse_shape(e[2][0]) if 2 in e else ()
    shard, offset, size = e.get(3, [0])[0], e.get(4, [0])[0], e.get(5, [0])[0]
    if shard not in shards:
      shards[shard] = np.memmap(_data_path(prefix, shard, num_shards), dtype=np.uint8, mode='r')
    chunk = np.asarray(shards[shard][offset:offset + size])
    if chunk.size != size:
      raise ValueError('%s: data shard truncated' % name)
    if verify_data_crc and 6 in e and mask_crc(crc32c(chunk.tobytes())) != e[6][0]:
      raise ValueError('%s: tensor checksum mismatch' % name)
    if dtype == DT_BFLOAT16:
      arr = (chunk.view('<u2').astype(np.uint32) << 16).view(np.float32)
    elif dtype in DT:
      arr = chunk.view(np.dtype(DT[dtype]).newbyteorder('<'))
    else:
      continue                                          # strings / resources: nothing the path needs
    out[name] = np.array(arr, copy=True).reshape(shape)
  return out


def _emit(field: int, wire: int, payload: bytes) -> bytes:
  return _put_varint((field << 3) | wire) + payload


def _entry_proto(arr: np.ndarray, offset: int, crc: int) -> bytes:
  shape = b''.join(_emit(2, 2, _put_varint(len(d)) + d) for d in
                   (_emit(1, 0, _put_varint(int(s))) for s in arr.shape))
  out = _emit(1, 0, _put_varint(DT_OF[arr.dtype]))
  out += _emit(2, 2, _put_varint(len(shape)) + shape)
  if offset:
    out += _emit(4, 0, _put_varint(offset))
  out += _emit(5, 0, _put_varint(arr.nbytes))
  out += _emit(6, 5, struct.pack('<I', crc))
  return out


def _build_block(entries: List[Tuple[bytes, bytes]]) -> bytes:
  """Restart interval 1 (no key prefix compression): every entry is a restart point."""
  body, restarts = bytearray(), []
  for key, value in entries:
    restarts.append(len(body))
    body += _put_varint(0) + _put_varint(len(key)) + _put_varint(len(value)) + key + value
  if not restarts:
    restarts = [0]
  for r in restarts:
    body += struct.pack('<I', r)
  body += struct.pack('<I', len(restarts))
  return bytes(body)


def write_bundle(values: Dict[str, np.ndarray], prefix: str, block_entries: int = 64) -> None:
  """Write `<prefix>.index` + `<prefix>.data-00000-of-00001` (one shard, little-endian, version 1)."""
  d = os.path.dirname(prefix)
  if d:
    os.makedirs(d, exist_ok=True)
  names = sorted(values, key=lambda s: s.encode('utf-8'))
  items: List[Tuple[bytes, bytes]] = []
  offset = 0
  with open(_data_path(prefix, 0, 1), 'wb') as f:
    for name in names:
      arr = np.asarray(values[name])
      if arr.ndim and not arr.flags.c_contiguous:
        arr = np.ascontiguousarray(arr)            # (ascontiguousarray would promote a 0-d scalar to 1-d)
      if arr.dtype not in DT_OF:
        raise TypeError('%s: dtype %s cannot be stored in a bundle' % (name, arr.dtype))
      raw = arr.astype(arr.dtype.newbyteorder('<'), copy=False).tobytes()
      f.write(raw)
      items.append((name.encode('utf-8'), _entry_proto(arr, offset, mask_crc(crc32c(raw)))))
      offset += len(raw)
  # BundleHeaderProto {num_shards = 1; endianness = LITTLE (0, default); version {producer = 1}}
  header = _emit(1, 0, _put_varint(1)) + _emit(3, 2, _put_varint(2) + _emit(1, 0, _put_varint(1)))
  items.insert(0, (b'', header))
  out = bytearray()
  index_entries: List[Tuple[bytes, bytes]] = []

  def put_block(block: bytes) -> bytes:
    off = len(out)
    out.extend(block)
    out.extend(b'\x00' + struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
    return _put_varint(off) + _put_varint(len(block))

  for i in range(0, len(items), block_entries):
    chunk = items[i:i + block_entries]
    index_entries.append((chunk[-1][0], put_block(_build_block(chunk))))
  meta_handle = put_block(_build_block([]))
  index_handle = put_block(_build_block(index_entries))
  footer = meta_handle + index_handle
  footer += b'\x00' * (FOOTER_LEN - 8 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
  out.extend(footer)
  with open(prefix + '.index', 'wb') as f:
    f.write(bytes(out))
