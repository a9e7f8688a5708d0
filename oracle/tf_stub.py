"""NumPy-eager stand-in for the few TensorFlow-1.x ops the reference's hot-path functions call
--  TEST INFRASTRUCTURE ONLY.

TensorFlow 1.x cannot be installed in this image (no wheel for Python 3.10, no network), so the
reference's learners cannot be imported as they are.  Their hot-path FUNCTIONS, however, are short
chains of stock TF ops.  `tests/golden/make_reference_golden.py` lifts those
functions out of the files under /root/reference with `ast` (no source is copied into this repo),
executes them with `tf` bound to this module, and stores inputs + outputs as fixtures.  That pins
oracle/pf_oracle.py against the reference's OWN Python code -- the op order, the bucket reshapes, the
`1e-10` / `1e-6` constants, which quantity is divided by which -- while the semantics of each
individual TF primitive remain a restatement of TF's published behaviour [3P]:

  * every op returns a float32 (or the input's integer) ndarray: one rounding per op, no fusion;
  * tf.round                     -> round-half-to-even (np.rint);
  * int / int                    -> float64 (tf.truediv on int32/int64 casts to float64);
  * tf.argmin                    -> first index on ties (np.argmin);
  * tf.contrib.distributions.percentile -> TF 1.12 sample_stats.percentile, interpolation='nearest':
      q, d in float64; descending sort (top_k); index = round_half_even((d-1) * (1 - q/100)) clipped;
  * tf.losses.softmax_cross_entropy -> mean over the batch of -sum(labels * log_softmax(logits));
  * tf.train.piecewise_constant  -> values[0] for x <= b[0], values[i] for b[i-1] < x <= b[i], ...
  * tf.reduce_sum                -> np.sum in the operand's dtype (NumPy's pairwise order, not Eigen's), axis list / keepdims honoured;
    tf.sqrt -> np.sqrt (correctly rounded) -- the proximal step of chn-pruned-gpu, tests/golden/make_reference_cpg_golden.py;
  * gradient_override_map / variable_scope / summaries -> no-ops (forward values only: gradients of
    the reference chain are NOT produced by this stub; the STE rules are pinned by hand-derived
    known answers in tests/test_oracle_kat.py).

A second group of stand-ins (tf.layers.conv2d / batch_normalization / dense / max_pooling2d, tf.pad,
tf.squeeze) lets the reference's NETWORK DEFINITIONS (utils/external/resnet_model.py) execute as they are,
with variables supplied by name (`variable_values`), so that the creation order, the auto-generated layer
names, the fixed-padding rule, the projection-shortcut placement and the BN constants of the restated
networks are pinned against the reference's own code.  The dense contractions use torch-CPU float32.

Only what the lifted functions use is implemented; anything else raises AttributeError loudly.
"""
from __future__ import annotations

import contextlib
import sys
import types

import numpy as np

float32, float64, int32, int64 = np.float32, np.float64, np.int32, np.int64
builtins_range, builtins_slice = range, slice


class _Dim(object):
  def __init__(self, v):
    self.value = None if v is None else int(v)

  def __int__(self):
    return self.value

  def __eq__(self, other):
    return self.value == (other.value if isinstance(other, _Dim) else other)


class TensorShape(object):
  ndims = property(lambda self: len(self.dims))

  def __init__(self, dims):
    self.dims = [_Dim(d) for d in dims]

  def __getitem__(self, i):
    return self.dims[i]

  def __len__(self):
    return len(self.dims)

  def as_list(self):
    return [d.value for d in self.dims]


def _raw(x):
  if isinstance(x, T):
    return x.a
  if isinstance(x, (list, tuple)) and any(isinstance(e, T) for e in x):
    return [np.asarray(_raw(e)) for e in x]
  return x


class T(object):
  """An eager tensor: a NumPy array plus the operator surface the lifted functions rely on."""
  __array_priority__ = 100

  def __init__(self, a, name='Tensor:0'):
    self.a = np.asarray(_raw(a))
    self.name = name

  # -- static shape API ---------------------------------------------------------------------------
  def set_shape(self, shape):
    assert tuple(int(d) for d in shape) == tuple(self.a.shape), (shape, self.a.shape)

  def get_shape(self):
    return TensorShape(self.a.shape)

  @property
  def shape(self):
    return TensorShape(self.a.shape)

  @property
  def dtype(self):
    return self.a.dtype.type

  def numpy(self):
    return self.a

  # -- arithmetic: NumPy's own float32 op == one correctly rounded float32 op -------------------------
  @staticmethod
  def _b(x, like):
    """Python scalars adopt the tensor's dtype, as tf.convert_to_tensor(..., dtype_hint) does."""
    x = _raw(x)
    if isinstance(x, (int, float)) and not isinstance(x, (bool, np.generic)):
      if np.issubdtype(like.dtype, np.floating):
        return like.dtype.type(x)
      if isinstance(x, int):
        return like.dtype.type(x)
    return x

  def __add__(self, o): return T(self.a + T._b(o, self.a))
  def __radd__(self, o): return T(T._b(o, self.a) + self.a)
  def __sub__(self, o): return T(self.a - T._b(o, self.a))
  def __rsub__(self, o): return T(T._b(o, self.a) - self.a)
  def __mul__(self, o): return T(self.a * T._b(o, self.a))
  def __rmul__(self, o): return T(T._b(o, self.a) * self.a)
  def __neg__(self): return T(-self.a)

  def __truediv__(self, o):
    b = T._b(o, self.a)
    if np.issubdtype(self.a.dtype, np.integer) and np.issubdtype(np.asarray(b).dtype, np.integer):
      return T(self.a.astype(np.float64) / np.asarray(b).astype(np.float64))     # tf.truediv on ints
    return T(self.a / b)

  def __rtruediv__(self, o):
    b = T._b(o, self.a)
    if np.issubdtype(self.a.dtype, np.integer) and np.issubdtype(np.asarray(b).dtype, np.integer):
      return T(np.asarray(b).astype(np.float64) / self.a.astype(np.float64))
    return T(b / self.a)

  def __floordiv__(self, o): return T(self.a // T._b(o, self.a))
  def __pow__(self, o): return T(self.a ** T._b(o, self.a))
  def __rpow__(self, o): return T(T._b(o, self.a) ** self.a)

  def __getitem__(self, idx):
    if isinstance(idx, tuple):
      idx = tuple(_raw(i) for i in idx)
    else:
      idx = _raw(idx)
    return T(self.a[idx])

  def __index__(self):
    return int(self.a)

  def __int__(self):
    return int(self.a)

  def __float__(self):
    return float(self.a)


def convert(x, dtype=None):
  a = np.asarray(_raw(x))
  if dtype is not None:
    a = a.astype(dtype)
  elif a.dtype == np.float64 and not isinstance(_raw(x), np.ndarray):
    a = a.astype(np.float32)             # Python floats become float32 constants
  return T(a)


# -- array ops ---------------------------------------------------------------------------------------

def constant(value, dtype=None, **kw): return convert(value, dtype)
uint8 = np.uint8
def cast(x, dtype, **kw): return T(np.asarray(_raw(x)).astype(dtype))
def reshape(x, shape, **kw): return T(np.reshape(_raw(x), [int(_raw(s)) for s in (shape.as_list() if isinstance(shape, TensorShape) else shape)]))
def ones(shape, dtype=np.float32, **kw): return T(np.ones(int(shape) if np.isscalar(shape) else [int(_raw(s)) for s in shape], dtype=dtype))
def zeros(shape, dtype=np.float32, **kw): return T(np.zeros(shape, dtype=dtype))
def concat(values, axis=0, **kw): return T(np.concatenate([np.atleast_1d(np.asarray(_raw(v))) for v in values], axis=axis))
def _py_default(x):
  """Python numbers / lists become float32 / int32 tensors (tf.convert_to_tensor defaults)."""
  if isinstance(x, (T, np.ndarray, np.generic)):
    return _raw(x)
  a = np.asarray(_raw(x))
  return a.astype(np.float32) if a.dtype == np.float64 else (a.astype(np.int32) if a.dtype == np.int64 else a)


def expand_dims(x, axis, **kw): return T(np.expand_dims(_py_default(x), int(axis)))
def tile(x, multiples, **kw): return T(np.tile(_raw(x), [int(m) for m in np.asarray(_raw(multiples)).reshape(-1)]))
def transpose(x, perm=None, **kw): return T(np.transpose(_raw(x), perm))
def gather(params, indices, axis=0, **kw): return T(np.take(_raw(params), np.asarray(_raw(indices)), axis=axis))
def stop_gradient(x, **kw): return T(_raw(x))
def identity(x, name=None, **kw): return T(_raw(x))
def sign(x, **kw): return T(np.sign(_raw(x)))
def abs(x, **kw): return T(np.abs(_raw(x)))                                     # noqa: A001
def round(x, **kw): return T(np.rint(_raw(x)))                                  # noqa: A001
def square(x, **kw): return T(_raw(x) * _raw(x))
def argmin(x, axis=None, **kw): return T(np.argmin(_raw(x), axis=axis).astype(np.int64))
def reduce_max(x, axis=None, **kw): return T(np.max(_raw(x), axis=axis))
def reduce_min(x, axis=None, **kw): return T(np.min(_raw(x), axis=axis))
def reduce_mean(x, axis=None, keepdims=False, keep_dims=False, name=None, **kw):
  keepdims = keepdims or keep_dims
  return _reduce_mean(x, axis, keepdims)


def _reduce_mean(x, axis=None, keepdims=False): return T(np.mean(_raw(x), axis=tuple(axis) if isinstance(axis, (list, tuple)) else axis, dtype=np.float32, keepdims=keepdims))
def reduce_sum(x, axis=None, keepdims=False, keep_dims=False, **kw):
  return T(np.sum(_raw(x), axis=tuple(axis) if isinstance(axis, (list, tuple)) else axis, dtype=np.asarray(_raw(x)).dtype,
                  keepdims=bool(keepdims or keep_dims)))
def sqrt(x, **kw): return T(np.sqrt(np.asarray(_raw(x))))
def minimum(a, b, **kw): return T(np.minimum(T._b(a, np.asarray(_raw(b))), T._b(b, np.asarray(_raw(a)))))
def maximum(a, b, **kw): return T(np.maximum(T._b(a, np.asarray(_raw(b))), T._b(b, np.asarray(_raw(a)))))
def pow(x, y, **kw): return T(np.power(_raw(x), T._b(y, np.asarray(_raw(x)))))  # noqa: A001
def add_n(xs, **kw):
  out = np.asarray(_raw(xs[0]))
  for x in xs[1:]:
    out = out + np.asarray(_raw(x))
  return T(out)


def argmax(x, axis=None, **kw): return T(np.argmax(_raw(x), axis=axis).astype(np.int64))
def equal(a, b, **kw): return T(np.equal(_raw(a), _raw(b)))
def where(c, x, y, **kw): return T(np.where(_raw(c), _raw(x), _raw(y)))
def range(*args, **kw):                                                         # noqa: A001
  vals = [np.asarray(_raw(a)) for a in args]
  dt = np.result_type(*[v.dtype for v in vals]) if vals else np.int32
  return T(np.arange(*[int(v) for v in vals]).astype(dt if np.issubdtype(dt, np.integer) else np.int32))
def linspace(start, stop, num, **kw): return T(np.linspace(start, stop, int(_raw(num))).astype(np.float32))


def map_fn(fn, elems, dtype=None, **kw):
  out = [np.asarray(_raw(fn(T(e)))) for e in np.asarray(_raw(elems))]
  a = np.stack(out, axis=0)
  return T(a.astype(dtype) if dtype is not None else a)


_scopes = []


class _ScopeObj(str):
  """What `with tf.variable_scope(...) as scope` yields: the absolute scope path; passing it back to
  tf.variable_scope re-enters that scope instead of nesting."""

  def reuse_variables(self):
    """Variables are looked up by name in `variable_values`, so reuse needs nothing."""


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, values=None, *a, **kw):
  saved = list(_scopes)
  if isinstance(name_or_scope, _ScopeObj):
    _scopes[:] = [p for p in str(name_or_scope).split('/') if p]
  else:
    name = name_or_scope if name_or_scope is not None else default_name
    _scopes.append(str(name))
  path = '/'.join(_scopes)
  try:
    yield _ScopeObj(path)
  finally:
    _scopes[:] = saved
    # TF closes the sub-scope counters of a variable scope on exit (VariableScopeStore.close_variable_subscopes),
    # so re-entering `path` names its layers dense, dense_1, ... again -- which is what makes `reuse` work
    for key in [k for k in _layer_names if k.startswith(path + '/')]:
      del _layer_names[key]


class _Scope(object):
  @property
  def name(self):
    return '/'.join(_scopes)


def get_variable_scope(): return _Scope()


created_variables = {}


def get_variable(name, shape=None, dtype=None, initializer=None, **kw):
  """The lifted functions only create `clusters` (initializer = a tensor)."""
  full = '/'.join(_scopes + [name])
  v = T(np.array(_raw(initializer), copy=True), name=full + ':0')
  created_variables[full] = v
  return v


class _Graph(object):
  @contextlib.contextmanager
  def gradient_override_map(self, m):
    yield


class Session(object):
  graph = _Graph()

  def __init__(self, *a, **kw):
    pass

  def run(self, fetches, feed_dict=None):
    """Eager stub: tensors already hold their values."""
    if isinstance(fetches, (list, tuple)):
      return [self.run(f) for f in fetches]
    return np.asarray(_raw(fetches))


def get_default_graph(): return _Graph()


# -- tf.nn / tf.losses / tf.train / tf.summary / tf.contrib ---------------------------------------------

def _softmax(z):
  z = np.asarray(_raw(z), dtype=np.float32)
  m = z.max(axis=-1, keepdims=True)
  e = np.exp((z - m).astype(np.float32))
  return (e / np.sum(e, axis=-1, keepdims=True, dtype=np.float32)).astype(np.float32)


def _log_softmax(z):
  z = np.asarray(_raw(z), dtype=np.float32)
  m = z.max(axis=-1, keepdims=True)
  s = (z - m).astype(np.float32)
  return (s - np.log(np.sum(np.exp(s), axis=-1, keepdims=True, dtype=np.float32))).astype(np.float32)


def _softmax_cross_entropy(onehot_labels, logits, weights=1.0, **kw):
  lab = np.asarray(_raw(onehot_labels), dtype=np.float32)
  per = -np.sum((lab * _log_softmax(logits)).astype(np.float32), axis=-1, dtype=np.float32)
  return T(np.float32(np.sum(per, dtype=np.float32) / np.float32(per.shape[0])))   # SUM_BY_NONZERO_WEIGHTS


def _piecewise_constant(x, boundaries, values, **kw):
  x = int(_raw(x))
  if x <= boundaries[0]:
    return T(np.float32(values[0]))
  for lo, hi, v in zip(boundaries[:-1], boundaries[1:], values[1:-1]):
    if lo < x <= hi:
      return T(np.float32(v))
  return T(np.float32(values[-1]))


def _exponential_decay(lr, global_step, decay_steps, decay_rate, staircase=False, **kw):
  p = np.float32(int(_raw(global_step))) / np.float32(decay_steps)
  if staircase:
    p = np.floor(p)
  return T(np.float32(np.float32(lr) * np.power(np.float32(decay_rate), np.float32(p))))


def _percentile(x, q, axis=None, interpolation=None, keep_dims=False, **kw):
  """tf.contrib.distributions.percentile, TF 1.12 sample_stats.py (default interpolation 'nearest')."""
  if interpolation not in (None, 'nearest'):
    raise NotImplementedError(interpolation)
  x = np.asarray(_raw(x))
  q = np.float64(np.asarray(_raw(q)))
  y = x.reshape(-1) if axis is None else np.moveaxis(x, axis, -1)
  d = np.float64(y.shape[-1])
  frac_at_q_or_above = np.float64(1.0) - q / np.float64(100.0)
  sorted_y = -np.sort(-y, axis=-1, kind='stable')
  idx = int(np.clip(np.int32(np.rint((d - 1.0) * frac_at_q_or_above)), 0, int(d) - 1))
  return T(sorted_y[..., idx])



# -- tf.layers subset (network definitions) ----------------------------------------------------------------
float16 = np.float16
variable_values = {}          # full variable name (without ':0') -> ndarray in the reference layout
variables_used = []           # creation order of the variables the executed network asked for
_layer_names = {}


def reset_layers(values):
  variable_values.clear()
  variable_values.update(values)
  del variables_used[:]
  _layer_names.clear()


def _layer_scope(base):
  """tf.layers auto-naming: `<base>`, `<base>_1`, ... unique per enclosing variable scope."""
  key = '/'.join(_scopes + [base])
  k = _layer_names.get(key, 0)
  _layer_names[key] = k + 1
  return base if k == 0 else '%s_%d' % (base, k)


def _var(layer, name):
  full = '/'.join(_scopes + [layer, name])
  variables_used.append(full)
  return np.asarray(variable_values[full], dtype=np.float32)


def _same_pad(size, k, s):
  out = -(-size // s)
  tot = max((out - 1) * s + k - size, 0)
  return tot // 2, tot - tot // 2


def _conv2d(inputs, filters, kernel_size, strides=1, padding='valid', use_bias=True, data_format='channels_last',
            kernel_initializer=None, **kw):
  import torch
  import torch.nn.functional as F
  assert data_format == 'channels_last'
  layer = kw.get('name') or _layer_scope('conv2d')
  if isinstance(kernel_size, (list, tuple)):
    kernel_size = kernel_size[0]
  w = _var(layer, 'kernel')                                        # HWIO
  x = torch.from_numpy(np.ascontiguousarray(_raw(inputs), dtype=np.float32)).permute(0, 3, 1, 2)
  k = w.shape[0]
  assert w.shape[0] == kernel_size and w.shape[3] == filters
  if str(padding).upper() == 'SAME':
    ph, pw = _same_pad(x.shape[2], k, strides), _same_pad(x.shape[3], k, strides)
    x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]))
  b = torch.from_numpy(_var(layer, 'bias')) if use_bias else None
  y = F.conv2d(x, torch.from_numpy(w).permute(3, 2, 0, 1).contiguous(), b, stride=strides)
  return T(y.permute(0, 2, 3, 1).contiguous().numpy())


def _batch_normalization(inputs, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, training=False,
                         fused=None, **kw):
  layer = _layer_scope('batch_normalization')
  x = np.asarray(_raw(inputs), dtype=np.float32)
  assert axis in (3, -1)
  gamma, beta = _var(layer, 'gamma'), _var(layer, 'beta')
  mm, mv = _var(layer, 'moving_mean'), _var(layer, 'moving_variance')
  if training:
    xr = x.reshape(-1, x.shape[-1]).astype(np.float64)
    mean, var = xr.mean(0), xr.var(0)
  else:
    mean, var = mm.astype(np.float64), mv.astype(np.float64)
  y = (x.astype(np.float64) - mean) / np.sqrt(var + epsilon) * gamma + beta
  return T(y.astype(np.float32))


def _dense(inputs, units, **kw):
  layer = kw.get('name') or _layer_scope('dense')
  w, b = _var(layer, 'kernel'), _var(layer, 'bias')
  assert w.shape[1] == units
  return T((np.asarray(_raw(inputs), np.float32) @ w + b).astype(np.float32))


def _layer_norm(inputs, center=True, scale=True, begin_norm_axis=1, begin_params_axis=-1, scope=None, **kw):
  """tf.contrib.layers.layer_norm [3P]: moments over axes >= begin_norm_axis, then
  tf.nn.batch_normalization(x, mean, var, beta, gamma, variance_epsilon=1e-12) = x * inv + (beta - mean * inv),
  inv = rsqrt(var + eps) * gamma; variables `LayerNorm[_k]/{beta,gamma}` over the last axis."""
  layer = scope or _layer_scope('LayerNorm')
  x = np.asarray(_raw(inputs), np.float32)
  axes = tuple(np.arange(x.ndim)[begin_norm_axis:])
  mean = np.mean(x, axis=axes, keepdims=True, dtype=np.float32)
  var = np.mean(np.square(x - mean), axis=axes, keepdims=True, dtype=np.float32)
  inv = (np.float32(1) / np.sqrt(var + np.float32(1e-12))).astype(np.float32)
  if scale:
    inv = inv * _var(layer, 'gamma')
  off = -mean * inv
  if center:
    off = _var(layer, 'beta') + off
  # creation order in TF is beta, gamma; keep `variables_used` in that order
  if scale and center:
    variables_used[-2], variables_used[-1] = variables_used[-1], variables_used[-2]
  return T((x * inv + off).astype(np.float32))


def sigmoid(x, **kw):
  x = np.asarray(_raw(x), np.float32)
  return T((np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32))


def shape(x, **kw): return T(np.array(np.shape(_raw(x)), dtype=np.int32))          # noqa: A001


# -- tf.image (utils/external/imagenet_preprocessing.py) ---------------------------------------------------------
image_hooks = {}     # 'window': (y, x, h, w) returned by sample_distorted_bounding_box; 'flip': bool


def _decode_jpeg(contents, channels=3, **kw):
  """libjpeg via Pillow (islow DCT, fancy upsampling: TF's defaults)."""
  import io
  from PIL import Image
  img = Image.open(io.BytesIO(bytes(_raw(contents).tobytes() if isinstance(_raw(contents), np.ndarray) else _raw(contents))))
  return T(np.asarray(img.convert('RGB' if channels == 3 else 'L'), dtype=np.uint8))


def _extract_jpeg_shape(contents, **kw):
  return T(np.array(_decode_jpeg(contents).a.shape, dtype=np.int32))


def _sample_distorted_bounding_box(image_size, bounding_boxes, **kw):
  y, x, h, w = image_hooks['window']
  return T(np.array([y, x, 0], np.int32)), T(np.array([h, w, -1], np.int32)), None


def _decode_and_crop_jpeg(contents, crop_window, channels=3, **kw):
  y, x, h, w = [int(v) for v in np.asarray(_raw(crop_window)).reshape(-1)]
  return T(_decode_jpeg(contents, channels).a[y:y + h, x:x + w])


def _random_flip_left_right(image, **kw):
  return T(_raw(image)[:, ::-1]) if image_hooks['flip'] else T(_raw(image))


def _resize_images(images, size, method=0, align_corners=False, **kw):
  """tf.image.resize_images(BILINEAR, align_corners=False) of TF 1.x [3P] (resize_bilinear_op.cc, legacy scaler):
  in = out * (in_size / (float) out_size); lo = floor(in); hi = min(ceil(in), in_size - 1); lerp = in - lo;
  out = top + (bottom - top) * ylerp with top = tl + (tr - tl) * xlerp; float32 throughout.  Written as plain loops
  on purpose (the oracle's vectorised version is checked against this one through the fixtures)."""
  assert method == 0 and not align_corners
  img = np.asarray(_raw(images)).astype(np.float32)
  ih, iw, ch = img.shape
  oh, ow = [int(_raw(v)) for v in size]
  sy, sx = np.float32(ih) / np.float32(oh), np.float32(iw) / np.float32(ow)
  out = np.zeros((oh, ow, ch), np.float32)
  for y in builtins_range(oh):
    iy = np.float32(y) * sy
    y0 = int(np.floor(iy)); y1 = min(int(np.ceil(iy)), ih - 1); ly = np.float32(iy - np.floor(iy))
    for x in builtins_range(ow):
      ix = np.float32(x) * sx
      x0 = int(np.floor(ix)); x1 = min(int(np.ceil(ix)), iw - 1); lx = np.float32(ix - np.floor(ix))
      top = img[y0, x0] + (img[y0, x1] - img[y0, x0]) * lx
      bot = img[y1, x0] + (img[y1, x1] - img[y1, x0]) * lx
      out[y, x] = top + (bot - top) * ly
  return T(out)


def decode_raw(bytes_, out_type, **kw):                                            # noqa: A002
  return T(np.frombuffer(bytes(_raw(bytes_)) if not isinstance(_raw(bytes_), np.ndarray) else _raw(bytes_).tobytes(), dtype=out_type))


def one_hot(indices, depth, **kw):
  idx = np.asarray(_raw(indices)).astype(np.int64)
  out = np.zeros(idx.shape + (int(depth),), np.float32)
  np.put_along_axis(out, idx[..., None], 1.0, axis=-1) if idx.ndim else out.__setitem__(int(idx), 1.0)
  return T(out)


def random_crop(value, size, **kw):
  """tf.random_crop with the offsets injected through image_hooks['crop'] = (oy, ox)."""
  oy, ox = image_hooks['crop']
  a = np.asarray(_raw(value))
  return T(a[oy:oy + int(size[0]), ox:ox + int(size[1]), :int(size[2])])


def _resize_image_with_crop_or_pad(image, target_height, target_width, **kw):
  """tf.image.resize_image_with_crop_or_pad [3P]: centred zero padding (extra pixel at the end) / centred crop."""
  a = np.asarray(_raw(image))
  h, w = a.shape[:2]
  out = a
  for axis, (cur, tgt) in enumerate(((h, int(target_height)), (w, int(target_width)))):
    if tgt > cur:
      lo = (tgt - cur) // 2
      pads = [(0, 0)] * out.ndim
      pads[axis] = (lo, tgt - cur - lo)
      out = np.pad(out, pads)
    elif tgt < cur:
      lo = (cur - tgt) // 2
      out = np.take(out, np.arange(lo, lo + tgt), axis=axis)
  return T(out)


def unstack(x, **kw): return [T(v) for v in np.asarray(_raw(x))]
def stack(values, axis=0, **kw): return T(np.stack([np.asarray(_raw(v)) for v in values], axis=axis))
def slice(x, begin, size, **kw):                                                   # noqa: A001
  a = np.asarray(_raw(x))
  idx = tuple(builtins_slice(int(_raw(b)), None if int(_raw(s)) < 0 else int(_raw(b)) + int(_raw(s))) for b, s in zip(begin, size))
  return T(a[idx])


def _max_pooling2d(inputs, pool_size, strides, padding='valid', data_format='channels_last', **kw):
  import torch
  import torch.nn.functional as F
  x = torch.from_numpy(np.ascontiguousarray(_raw(inputs), dtype=np.float32)).permute(0, 3, 1, 2)
  if isinstance(pool_size, (list, tuple)):
    pool_size = pool_size[0]
  if str(padding).upper() == 'SAME':
    ph, pw = _same_pad(x.shape[2], pool_size, strides), _same_pad(x.shape[3], pool_size, strides)
    x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]), value=float('-inf'))
  return T(F.max_pool2d(x, pool_size, strides).permute(0, 2, 3, 1).contiguous().numpy())


def _flatten(inputs, **kw):
  a = np.asarray(_raw(inputs))
  return T(a.reshape(a.shape[0], -1))


def pad(x, paddings, **kw): return T(np.pad(_raw(x), [tuple(p) for p in paddings]))
def squeeze(x, axis=None, name=None, **kw): return T(np.squeeze(_raw(x), axis=tuple(axis) if axis is not None else None))
def variance_scaling_initializer(*a, **kw): return 'variance_scaling'


# -- tf.contrib.slim subset (MobileNet-v1 definition) ---------------------------------------------------------
_arg_stack = [{}]


def _slim_fn(fn):
  def wrapped(*args, **kwargs):
    merged = dict(_arg_stack[-1].get(wrapped, {}))
    merged.update(kwargs)
    return fn(*args, **merged)
  wrapped.__name__ = fn.__name__
  return wrapped


@contextlib.contextmanager
def _arg_scope(fns_or_scope, **kwargs):
  if isinstance(fns_or_scope, dict):                 # re-entering a scope captured with `as sc`
    new = {k: dict(v) for k, v in fns_or_scope.items()}
  else:
    new = {k: dict(v) for k, v in _arg_stack[-1].items()}
    for f in fns_or_scope:
      new.setdefault(f, {}).update(kwargs)
  _arg_stack.append(new)
  try:
    yield new
  finally:
    _arg_stack.pop()


def _nhwc_conv(x, w_oihw, stride, padding, groups=1, bias=None):
  import torch
  import torch.nn.functional as F
  x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).permute(0, 3, 1, 2)
  k = w_oihw.shape[2]
  if str(padding).upper() == 'SAME':
    ph, pw = _same_pad(x.shape[2], k, stride), _same_pad(x.shape[3], k, stride)
    x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]))
  y = F.conv2d(x, w_oihw, bias, stride=stride, groups=groups)
  return y.permute(0, 2, 3, 1).contiguous().numpy()


@_slim_fn
def _slim_batch_norm(inputs, decay=0.999, center=False, scale=False, epsilon=0.001, is_training=True,
                     updates_collections=None, scope=None, **kw):
  with variable_scope(scope, 'BatchNorm'):
    x = np.asarray(_raw(inputs), dtype=np.float32)
    C = x.shape[-1]
    beta = _var_here('beta') if center else np.zeros(C, np.float32)
    gamma = _var_here('gamma') if scale else np.ones(C, np.float32)
    mm, mv = _var_here('moving_mean'), _var_here('moving_variance')
    if is_training:
      xr = x.reshape(-1, C).astype(np.float64)
      mean, var = xr.mean(0), xr.var(0)
    else:
      mean, var = mm.astype(np.float64), mv.astype(np.float64)
    return T(((x.astype(np.float64) - mean) / np.sqrt(var + epsilon) * gamma + beta).astype(np.float32))


def _var_here(name):
  full = '/'.join(_scopes + [name])
  variables_used.append(full)
  return np.asarray(variable_values[full], dtype=np.float32)


def _post(y, normalizer_fn, normalizer_params, activation_fn):
  out = T(y)
  if normalizer_fn is not None:
    out = normalizer_fn(out, **(normalizer_params or {}))
  if activation_fn is not None:
    out = activation_fn(out)
  return out


@_slim_fn
def _slim_conv2d(inputs, num_outputs, kernel_size, stride=1, padding='SAME', rate=1, activation_fn=None,
                 normalizer_fn=None, normalizer_params=None, weights_initializer=None, weights_regularizer=None,
                 scope=None, **kw):
  import torch
  with variable_scope(scope, 'Conv'):
    w = _var_here('weights')                                           # HWIO
    assert list(w.shape[:2]) == list(kernel_size) and w.shape[3] == num_outputs and rate == 1
    bias = torch.from_numpy(_var_here('biases')) if normalizer_fn is None else None
    y = _nhwc_conv(_raw(inputs), torch.from_numpy(w).permute(3, 2, 0, 1).contiguous(), stride, padding, 1, bias)
    return _post(y, normalizer_fn, normalizer_params, activation_fn)


@_slim_fn
def _slim_separable_conv2d(inputs, num_outputs, kernel_size, depth_multiplier=1, stride=1, padding='SAME', rate=1,
                           activation_fn=None, normalizer_fn=None, normalizer_params=None, weights_initializer=None,
                           weights_regularizer=None, scope=None, **kw):
  import torch
  assert num_outputs is None and depth_multiplier == 1 and rate == 1      # depthwise only (mobilenet_v1.py:279-284)
  with variable_scope(scope, 'SeparableConv2d'):
    w = _var_here('depthwise_weights')                                 # [kh, kw, C, 1]
    C = w.shape[2]
    y = _nhwc_conv(_raw(inputs), torch.from_numpy(w).permute(2, 3, 0, 1).contiguous(), stride, padding, C)
    return _post(y, normalizer_fn, normalizer_params, activation_fn)


@_slim_fn
def _slim_dropout(inputs, keep_prob=0.5, is_training=True, scope=None, **kw):
  """Deterministic stand-in: in training mode NOTHING is dropped (the overwhelmingly likely outcome at
  keep_prob 0.999) and the survivors are scaled by 1/keep_prob, as tf.nn.dropout does."""
  x = np.asarray(_raw(inputs), dtype=np.float32)
  return T((x / np.float32(keep_prob)).astype(np.float32) if is_training else x)


@_slim_fn
def _slim_avg_pool2d(inputs, kernel_size, stride=2, padding='VALID', scope=None, **kw):
  import torch
  import torch.nn.functional as F
  x = torch.from_numpy(np.ascontiguousarray(_raw(inputs), dtype=np.float32)).permute(0, 3, 1, 2)
  assert str(padding).upper() == 'VALID'
  return T(F.avg_pool2d(x, tuple(kernel_size), stride).permute(0, 2, 3, 1).contiguous().numpy())


class _Flags(object):
  """tf.app.flags: DEFINE_* + a FLAGS namespace the generator script fills in."""

  class _Values(object):
    pass

  def __init__(self):
    self.FLAGS = _Flags._Values()

  def _define(self, name, default, help=None):        # noqa: A002
    if not hasattr(self.FLAGS, name):
      setattr(self.FLAGS, name, default)
  DEFINE_integer = DEFINE_float = DEFINE_string = DEFINE_boolean = DEFINE_bool = \
      lambda self, name, default, help=None: self._define(name, default, help)      # noqa: A002


def _ns(name, **attrs):
  m = types.ModuleType(name)
  for k, v in attrs.items():
    setattr(m, k, v)
  return m


def install() -> types.ModuleType:
  """Register this stub as `tensorflow` (+ the sub-modules the reference imports) in sys.modules."""
  this = sys.modules[__name__]
  tf = _ns('tensorflow')
  for k in dir(this):
    if not k.startswith('_') and k not in ('install', 'sys', 'types', 'np', 'contextlib', 'annotations', 'builtins_range', 'builtins_slice', 'image_hooks'):
      setattr(tf, k, getattr(this, k))
  flags = _Flags()
  tf.app = _ns('tensorflow.app', flags=flags, run=lambda *a, **k: None)
  tf.nn = _ns('tensorflow.nn', softmax=lambda z, name=None, **kw: T(_softmax(z)), relu=lambda x, name=None, **kw: T(np.maximum(_raw(x), np.float32(0))),
              relu6=lambda x, name=None, **kw: T(np.minimum(np.maximum(_raw(x), np.float32(0)), np.float32(6))))
  tf.losses = _ns('tensorflow.losses', softmax_cross_entropy=_softmax_cross_entropy)
  tf.nn.l2_loss = lambda v, **kw: T(np.float32(np.sum(np.square(np.asarray(_raw(v), np.float32)), dtype=np.float32) / np.float32(2)))

  def _in_top_k(predictions, targets, k, **kw):
    p = np.asarray(_raw(predictions), np.float32)
    t = np.asarray(_raw(targets)).astype(np.int64)
    tv = p[np.arange(p.shape[0]), t][:, None]
    return T(np.sum(p > tv, axis=1) < k)            # ties count in favour (TF in_top_k)
  tf.nn.in_top_k = _in_top_k
  tf.train = _ns('tensorflow.train', piecewise_constant=_piecewise_constant, exponential_decay=_exponential_decay)
  tf.summary = _ns('tensorflow.summary', scalar=lambda *a, **k: None)
  tf.layers = _ns('tensorflow.layers', conv2d=_conv2d, batch_normalization=_batch_normalization, dense=_dense,
                  max_pooling2d=_max_pooling2d, flatten=_flatten)
  tf.image = _ns('tensorflow.image', decode_jpeg=_decode_jpeg, extract_jpeg_shape=_extract_jpeg_shape,
                 sample_distorted_bounding_box=_sample_distorted_bounding_box, decode_and_crop_jpeg=_decode_and_crop_jpeg,
                 random_flip_left_right=_random_flip_left_right, resize_images=_resize_images,
                 resize_image_with_crop_or_pad=_resize_image_with_crop_or_pad,
                 ResizeMethod=_ns('ResizeMethod', BILINEAR=0))
  tf.test = _ns('tensorflow.test', is_built_with_cuda=lambda: False)
  tf.logging = _ns('tensorflow.logging', info=lambda *a, **k: None, warning=lambda *a, **k: None, debug=lambda *a, **k: None,
                   error=lambda *a, **k: None)
  dist = _ns('tensorflow.contrib.distributions', percentile=_percentile)
  ge = _ns('tensorflow.contrib.graph_editor')
  slim = _ns('tensorflow.contrib.slim', arg_scope=_arg_scope, conv2d=_slim_conv2d, separable_conv2d=_slim_separable_conv2d,
             batch_norm=_slim_batch_norm, dropout=_slim_dropout, avg_pool2d=_slim_avg_pool2d)
  clayers = _ns('tensorflow.contrib.layers', softmax=lambda logits, scope=None: T(_softmax(logits)),
                l2_regularizer=lambda wd: ('l2', wd), layer_norm=_layer_norm)
  tf.contrib = _ns('tensorflow.contrib', distributions=dist, graph_editor=ge, slim=slim, layers=clayers)
  tf.truncated_normal_initializer = lambda stddev=1.0, **kw: ('truncated_normal', stddev)
  tf.GraphKeys = _ns('tensorflow.GraphKeys', UPDATE_OPS='update_ops')
  sys.modules.update({'tensorflow': tf, 'tensorflow.contrib': tf.contrib,
                      'tensorflow.contrib.graph_editor': ge, 'tensorflow.contrib.distributions': dist})
  return tf
