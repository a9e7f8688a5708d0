"""An eager TF stand-in WITH gradients (torch autograd + TF's gradient_override_map)  --  TEST INFRASTRUCTURE ONLY.

oracle/tf_stub.py runs the reference's quantiser code over NumPy and pins its FORWARD values.  The backward rules of those
quantisers are not ordinary derivatives: the reference rewires TF's gradient registry while it builds the graph,

    with g.gradient_override_map({'Round': 'Identity'}):                  # uq utils.py:185, nuq utils.py:268
    with g.gradient_override_map({'Mul': 'Add', 'Sign': 'Identity'}):    # nuq utils.py:305, 345

and cuts the range statistics out with tf.stop_gradient (uq utils.py:224-225).  This module executes the same functions
(lifted from /root/reference by tests/golden/make_reference_golden.py) on torch tensors so that `tf.gradients` becomes
torch.autograd.grad, with the override map honoured op by op:
    Round under 'Identity'  -> the upstream gradient passes unchanged;
    Mul   under 'Add'       -> BOTH factors receive the upstream gradient (un-broadcast by summation), as AddGrad does;
    Sign  under 'Identity'  -> the upstream gradient passes unchanged (instead of zeros).
Every other op has its ordinary derivative; reduce_max / reduce_min only ever appear under stop_gradient.
What this pins: the gradients the reference's graph code DEFINES (which operands get what), i.e. the straight-through rules
that oracle/pf_oracle.py states by hand (uniform_quantize_grad, nuq_backward).  What it does not pin: TensorFlow's kernels.
Only tests/ and tests/golden/ may import this module.
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch

from oracle import tf_stub as _np_stub

float32 = np.float32
int64 = np.int64
int32 = np.int32

_override = [{}]              # stack of gradient_override_map dictionaries


def _t(x, dtype=None):
  if isinstance(x, T):
    return x.v
  if isinstance(x, torch.Tensor):
    return x
  if isinstance(x, (list, tuple)) and any(isinstance(e, (T, torch.Tensor)) for e in x):
    return torch.stack([_t(e) for e in x])
  a = np.asarray(x)
  if dtype is None:
    dtype = np.float32 if a.dtype.kind == 'f' else (np.int64 if a.dtype.kind in 'iu' else a.dtype)
  return torch.as_tensor(a.astype(dtype))


class _Dim(object):
  def __init__(self, v):
    self.value = int(v)

  def __int__(self):
    return self.value

  def __index__(self):
    return self.value


class _Shape(object):
  def __init__(self, dims):
    self.dims = [int(d) for d in dims]

  def __getitem__(self, i):
    return _Dim(self.dims[i])

  def __len__(self):
    return len(self.dims)

  def as_list(self):
    return list(self.dims)


class T(object):
  """A tensor: wraps a torch tensor; arithmetic follows the active gradient_override_map."""

  def __init__(self, v):
    self.v = v if isinstance(v, torch.Tensor) else _t(v)

  def get_shape(self):
    return _Shape(self.v.shape)

  @property
  def shape(self):
    return _Shape(self.v.shape)

  def numpy(self):
    return self.v.detach().numpy()

  def __add__(self, o): return T(self.v + _t(o))
  def __radd__(self, o): return T(_t(o) + self.v)
  def __sub__(self, o): return T(self.v - _t(o))
  def __rsub__(self, o): return T(_t(o) - self.v)
  def __mul__(self, o): return _mul(self, o)
  def __rmul__(self, o): return _mul(o, self)
  def __truediv__(self, o): return T(self.v / _t(o))
  def __rtruediv__(self, o): return T(_t(o) / self.v)
  def __neg__(self): return T(-self.v)
  def __pow__(self, o): return T(self.v ** _t(o))
  def __rpow__(self, o): return T(_t(o) ** self.v)

  def __getitem__(self, idx):
    if isinstance(idx, tuple):
      idx = tuple(int(i.v) if isinstance(i, T) else i for i in idx)
    elif isinstance(idx, T):
      idx = int(idx.v)
    return T(self.v[idx])

  def __int__(self): return int(self.v)
  def __index__(self): return int(self.v)
  def __float__(self): return float(self.v)


# ---- ops whose gradient the reference overrides ------------------------------------------------------------------------
def _unbroadcast(g, shape):
  while g.dim() > len(shape):
    g = g.sum(0)
  for i, s in enumerate(shape):
    if s == 1 and g.shape[i] != 1:
      g = g.sum(i, keepdim=True)
  return g


class _MulAsAdd(torch.autograd.Function):
  @staticmethod
  def forward(ctx, a, b):
    ctx.shapes = (tuple(a.shape), tuple(b.shape))
    return a * b

  @staticmethod
  def backward(ctx, g):                       # AddGrad: both inputs get the upstream gradient
    return _unbroadcast(g, ctx.shapes[0]), _unbroadcast(g, ctx.shapes[1])


class _PassThrough(torch.autograd.Function):
  """Forward = the given function, backward = Identity's gradient."""

  @staticmethod
  def forward(ctx, x, fn):
    return fn(x)

  @staticmethod
  def backward(ctx, g):
    return g, None


def _mul(a, b):
  ta, tb = _t(a), _t(b)
  if _override[-1].get('Mul') == 'Add' and ta.is_floating_point() and tb.is_floating_point():
    return T(_MulAsAdd.apply(ta, tb))
  return T(ta * tb)


def round(x, **kw):                                    # noqa: A001
  if _override[-1].get('Round') == 'Identity':
    return T(_PassThrough.apply(_t(x), torch.round))
  return T(torch.round(_t(x)))


def sign(x, **kw):
  if _override[-1].get('Sign') == 'Identity':
    return T(_PassThrough.apply(_t(x), torch.sign))
  return T(torch.sign(_t(x)))


class _Graph(object):
  @contextlib.contextmanager
  def gradient_override_map(self, m):
    _override.append(dict(_override[-1], **m))
    try:
      yield
    finally:
      _override.pop()


class Session(object):
  graph = _Graph()

  def __init__(self, *a, **kw):
    pass


def get_default_graph():
  return _Graph()


# ---- ordinary ops --------------------------------------------------------------------------------------------------------
def stop_gradient(x, **kw): return T(_t(x).detach())
def constant(value, dtype=None, **kw): return T(_t(value, dtype))
def identity(x, **kw): return T(_t(x))
def abs(x, **kw): return T(torch.abs(_t(x)))           # noqa: A001
def square(x, **kw): return T(_t(x) * _t(x))


def cast(x, dtype, **kw):
  v = _t(x)
  return T(v.to(torch.float32) if np.dtype(dtype).kind == 'f' else v.to(torch.int64))


def reduce_max(x, axis=None, **kw):
  v = _t(x)
  return T(v.max() if axis is None else v.max(dim=axis).values)


def reduce_min(x, axis=None, **kw):
  v = _t(x)
  return T(v.min() if axis is None else v.min(dim=axis).values)


def reduce_sum(x, axis=None, **kw):
  v = _t(x)
  return T(v.sum() if axis is None else v.sum(dim=axis))


def _ints(shape):
  if isinstance(shape, _Shape):
    return shape.as_list()
  if isinstance(shape, T):
    return [int(s) for s in shape.v.reshape(-1).tolist()]
  out = []
  for s in (shape if isinstance(shape, (list, tuple)) else [shape]):
    if isinstance(s, T):
      out.extend(int(i) for i in s.v.reshape(-1).tolist())
    else:
      out.append(int(s))
  return out


def reshape(x, shape, **kw): return T(_t(x).reshape(_ints(shape)))
def ones(shape, dtype=np.float32, **kw): return T(torch.ones(_ints(shape), dtype=torch.float32 if np.dtype(dtype).kind == 'f' else torch.int64))
def zeros(shape, dtype=np.float32, **kw): return T(torch.zeros(_ints(shape), dtype=torch.float32 if np.dtype(dtype).kind == 'f' else torch.int64))


def concat(values, axis=0, **kw):
  ts = [torch.atleast_1d(_t(v)) for v in values]
  if any(t.dtype == torch.int64 for t in ts):
    ts = [t.to(torch.int64) for t in ts]
  return T(torch.cat(ts, dim=axis))


def expand_dims(x, axis, **kw): return T(_t(x).unsqueeze(int(axis)))
def tile(x, multiples, **kw): return T(_t(x).repeat(*_ints(multiples)))
def transpose(x, perm=None, **kw): return T(_t(x).permute(*perm) if perm is not None else _t(x).t())
def argmin(x, axis=None, **kw): return T(torch.argmin(_t(x), dim=axis))          # first index on ties, like tf.argmin
def gather(params, indices, axis=0, **kw): return T(torch.index_select(_t(params), 0, _t(indices).reshape(-1)).reshape(tuple(_t(indices).shape) + tuple(_t(params).shape[1:])))
def range(*args, **kw): return T(torch.arange(*[int(_t(a)) for a in args], dtype=torch.int64))      # noqa: A001
def linspace(start, stop, num, **kw): return T(torch.as_tensor(np.linspace(start, stop, int(_t(num))).astype(np.float32)))
def stack(values, axis=0, **kw): return T(torch.stack([_t(v) for v in values], dim=axis))


def map_fn(fn, elems, dtype=None, **kw):
  return T(torch.stack([_t(fn(T(e))) for e in _t(elems)], dim=0))


# ---- scopes / variables ----------------------------------------------------------------------------------------------------
_scopes = []
created_variables = {}


@contextlib.contextmanager
def variable_scope(name, *a, **kw):
  _scopes.append(str(name))
  try:
    yield str(name)
  finally:
    _scopes.pop()


class _Scope(object):
  @property
  def name(self):
    return '/'.join(_scopes)


def get_variable_scope(): return _Scope()


def get_variable(name, shape=None, dtype=None, initializer=None, **kw):
  """The quantisers only create `clusters` (initializer = a tensor): a leaf that takes gradients."""
  full = '/'.join(_scopes + [name])
  v = T(_t(initializer).detach().clone().requires_grad_(True))
  created_variables[full] = v
  return v


# ---- tf.contrib.distributions.percentile: no gradient is taken through the codebook initialisation ------------------------
def _percentile(x, q, axis=None, **kw):
  out = _np_stub._percentile(_np_stub.T(_t(x).detach().numpy()), _np_stub.T(np.asarray(_t(q).detach().numpy() if isinstance(q, T) else q)), axis=axis)
  return T(torch.as_tensor(np.asarray(_np_stub._raw(out), dtype=np.float32)))


class _Distributions(object):
  percentile = staticmethod(_percentile)


class _Contrib(object):
  distributions = _Distributions()
  graph_editor = None


contrib = _Contrib()


def gradients(ys, xs, grad_ys=None):
  g = torch.autograd.grad(_t(ys), [_t(x) for x in xs], grad_outputs=None if grad_ys is None else _t(grad_ys), allow_unused=True)
  return [None if v is None else T(v) for v in g]
