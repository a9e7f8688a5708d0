"""A deferred-execution (TF-1.x graph + session) stand-in on torch  --  TEST INFRASTRUCTURE ONLY.

oracle/tf_stub.py executes the reference's *eager-able* functions over NumPy.  The reference's DDPG agent
(rl_agents/ddpg/agent.py:136-418) is graph code: placeholders, `optimizer.minimize`, `tf.assign`, `sess.run(fetches,
feed_dict)`.  This module provides just enough of that programming model to EXECUTE `Agent.__build / init / train` as they
are written, so that the agent's update step can be pinned by fixtures made from the reference's own code
(tests/golden/make_reference_rl_golden.py) instead of by a second restatement:

  * every tf call builds a `Node` (a closure over its input nodes); nothing is computed at build time;
  * `Session.run` evaluates the fetched nodes with the feed dict, memoising per call; all fetches of one call see the
    variable values from BEFORE the call and all assignments are applied after every fetch has been evaluated (TF gives
    no ordering between un-sequenced ops of one `run`; the agent's fetches -- losses and the two optimiser steps, which
    touch disjoint variables and read each other's variables only through the forward pass -- are order-independent
    under exactly this snapshot rule);
  * gradients come from torch autograd (float32 CPU); the primitives are restated from TF's documented semantics, as in
    oracle/tf_stub.py:  tf.layers.dense = x @ kernel + bias (glorot-uniform kernel, zero bias);
    tf.contrib.layers.layer_norm = moments over the last axis, variance_epsilon 1e-12, beta/gamma;
    tf.nn.l2_loss = sum(x^2)/2;  tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), m, v, eps = 1e-8 OUTSIDE the
    square root, slots and beta powers initialised by `variables_initializer(optimizer.variables())`;
    tf.random_normal draws from this module's seeded generator (`seed(n)`).
What it does NOT pin: TensorFlow's kernels themselves (never ran here) -- the same caveat as for every other fixture.
Only tests/ and tests/golden/ may import this module.
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch

float32 = torch.float32
_rng = np.random.RandomState(0)


def seed(n: int) -> None:
  global _rng
  _rng = np.random.RandomState(n)


# ---------------------------------------------------------------------------------------------------------------------
# graph nodes
# ---------------------------------------------------------------------------------------------------------------------
class Node(object):
  out_dim = None                # static width of the last axis where a layer constructor needs it (shape inference)

  def __init__(self, fn, inputs=(), name='op'):
    self.fn, self.inputs, self.name = fn, tuple(inputs), name

  # arithmetic builds nodes
  def __add__(self, o): return _binary(torch.add, self, o)
  def __radd__(self, o): return _binary(torch.add, o, self)
  def __sub__(self, o): return _binary(torch.sub, self, o)
  def __rsub__(self, o): return _binary(torch.sub, o, self)
  def __mul__(self, o): return _binary(torch.mul, self, o)
  def __rmul__(self, o): return _binary(torch.mul, o, self)
  def __truediv__(self, o): return _binary(torch.div, self, o)
  def __neg__(self): return Node(lambda x: -x, [self], 'neg')

  def assign(self, value):
    return assign(self, value)


class Placeholder(Node):
  def __init__(self, name):
    Node.__init__(self, None, (), name)


class Variable(Node):
  def __init__(self, name, shape, initializer, trainable):
    Node.__init__(self, None, (), name + ':0')
    self.shape_, self.initializer, self.trainable = tuple(int(s) for s in shape), initializer, trainable
    self.value = None                       # torch tensor once initialised

  def initial_value(self):
    return self.initializer(self.shape_)


class Assign(Node):
  """An op with a side effect: evaluated like any node (value = the new value), applied by the session afterwards."""

  def __init__(self, var, value):
    Node.__init__(self, None, [_as_node(value)], 'assign')
    self.var = var


class Group(Node):
  def __init__(self, ops):
    Node.__init__(self, None, list(ops), 'group')


def _as_node(x):
  if isinstance(x, Node):
    return x
  t = torch.as_tensor(np.asarray(x, dtype=np.float32))
  return Node(lambda: t, [], 'const')


def _binary(f, a, b):
  a, b = _as_node(a), _as_node(b)
  out = Node(lambda x, y: f(x, y), [a, b], f.__name__)
  out.out_dim = a.out_dim if a.out_dim is not None else b.out_dim
  return out


def _unary(f, name):
  def build(x, **kw):
    x = _as_node(x)
    out = Node(lambda v: f(v), [x], name)
    out.out_dim = x.out_dim
    return out
  return build


# ---------------------------------------------------------------------------------------------------------------------
# variables, scopes, collections
# ---------------------------------------------------------------------------------------------------------------------
class GraphKeys(object):
  GLOBAL_VARIABLES = 'variables'
  TRAINABLE_VARIABLES = 'trainable_variables'


_scopes = []
_layer_names = {}
_variables = {}               # full name -> Variable, creation order (dict order)


def reset_default_graph():
  del _scopes[:]
  _layer_names.clear()
  _variables.clear()


class _ScopeObj(str):
  def reuse_variables(self):
    """Variables are looked up by full name: re-entering a scope finds them."""


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, *a, **kw):
  saved = list(_scopes)
  if isinstance(name_or_scope, _ScopeObj):
    _scopes[:] = [p for p in str(name_or_scope).split('/') if p]
  else:
    _scopes.extend(p for p in str(name_or_scope if name_or_scope is not None else default_name).split('/') if p)
  path = '/'.join(_scopes)
  try:
    yield _ScopeObj(path)
  finally:
    _scopes[:] = saved
    # TF closes the sub-scope counters of a variable scope on exit, so re-entering `path` names its layers dense,
    # dense_1, ... again -- which is what makes `reuse` find the same variables (cf. oracle/tf_stub.py)
    for key in [k for k in _layer_names if k.startswith(path + '/')]:
      del _layer_names[key]


def _layer_scope(base):
  key = '/'.join(_scopes + [base])
  k = _layer_names.get(key, 0)
  _layer_names[key] = k + 1
  return base if k == 0 else '%s_%d' % (base, k)


def zeros_initializer():
  return lambda shape: np.zeros(shape, np.float32)


def ones_initializer():
  return lambda shape: np.ones(shape, np.float32)


def _glorot_uniform(shape):
  fan_in, fan_out = shape[0], shape[1]
  limit = np.sqrt(6.0 / (fan_in + fan_out))
  return _rng.uniform(-limit, limit, size=shape).astype(np.float32)


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **kw):
  full = '/'.join(_scopes + [name])
  if full in _variables:
    return _variables[full]
  init = initializer if initializer is not None else (_glorot_uniform if len(shape) == 2 else zeros_initializer())
  v = _variables[full] = Variable(full, shape, init, trainable)
  return v


def get_collection(key, scope=None):
  out = []
  for name, v in _variables.items():
    if scope is not None and not (name == scope or name.startswith(scope + '/')):
      continue
    if key == GraphKeys.TRAINABLE_VARIABLES and not v.trainable:
      continue
    out.append(v)
  return out


def placeholder(dtype, shape=None, name=None):
  p = Placeholder(name or 'placeholder')
  if shape:
    p.out_dim = shape[-1]
  return p


def _initializer_op(v):
  """ONE initialiser op per variable (TF's `var.initializer`): two groups that both initialise `v` in one run draw once."""
  if getattr(v, '_init_op', None) is None:
    v._init_op = Assign(v, Node(lambda v=v: torch.as_tensor(np.asarray(v.initial_value(), dtype=np.float32)), [], 'init'))
  return v._init_op


def variables_initializer(var_list):
  return Group([_initializer_op(v) for v in var_list])


def assign(var, value):
  return Assign(var, value)


def group(*ops):
  return Group(ops)


# ---------------------------------------------------------------------------------------------------------------------
# ops
# ---------------------------------------------------------------------------------------------------------------------
square = _unary(torch.square, 'square')
sqrt = _unary(torch.sqrt, 'sqrt')
abs = _unary(torch.abs, 'abs')                     # noqa: A001
sigmoid = _unary(torch.sigmoid, 'sigmoid')


def shape(x, **kw):
  return Node(lambda v: torch.tensor(list(v.shape), dtype=torch.int64), [_as_node(x)], 'shape')


def reduce_mean(x, axis=None, **kw):
  return Node(lambda v: v.mean() if axis is None else v.mean(dim=axis), [_as_node(x)], 'mean')


def reduce_sum(x, axis=None, **kw):
  return Node(lambda v: v.sum() if axis is None else v.sum(dim=axis), [_as_node(x)], 'sum')


def maximum(a, b, **kw): return _binary(torch.maximum, a, b)


def add_n(xs, **kw):
  xs = [_as_node(x) for x in xs]

  def f(*vs):
    out = vs[0]
    for v in vs[1:]:
      out = out + v
    return out
  return Node(f, xs, 'add_n')


def concat(values, axis=0, **kw):
  values = [_as_node(v) for v in values]
  out = Node(lambda *vs: torch.cat(vs, dim=axis), values, 'concat')
  if axis in (1, -1) and all(v.out_dim is not None for v in values):
    out.out_dim = sum(v.out_dim for v in values)
  return out


def clip_by_value(x, lo, hi, **kw):
  x = _as_node(x)
  out = Node(lambda v: torch.clamp(v, float(lo), float(hi)), [x], 'clip')
  out.out_dim = x.out_dim
  return out


def cast(x, dtype, **kw):
  return Node(lambda v: v.to(torch.float32), [_as_node(x)], 'cast')


def random_normal(shape_, mean=0.0, stddev=1.0, **kw):
  def f(s, sd):
    dims = [int(d) for d in s.reshape(-1).tolist()]
    return torch.as_tensor(_rng.standard_normal(dims).astype(np.float32)) * sd + float(mean)
  return Node(f, [_as_node(shape_), _as_node(stddev)], 'random_normal')


class _NN(object):
  relu = staticmethod(_unary(torch.relu, 'relu'))

  @staticmethod
  def l2_loss(x, **kw):
    return Node(lambda v: (v * v).sum() / 2, [_as_node(x)], 'l2_loss')


nn = _NN()


def _dense(inputs, units, **kw):
  layer = kw.get('name') or _layer_scope('dense')
  x = _as_node(inputs)
  with variable_scope(layer):
    w = get_variable('kernel', (x.out_dim, units), initializer=_glorot_uniform)
    b = get_variable('bias', (units,), initializer=zeros_initializer())
  out = Node(lambda v, wv, bv: v @ wv + bv, [x, w, b], 'dense')
  out.out_dim = units
  return out


def _layer_norm(inputs, **kw):
  layer = kw.get('scope') or _layer_scope('LayerNorm')
  x = _as_node(inputs)
  with variable_scope(layer):
    beta = get_variable('beta', (x.out_dim,), initializer=zeros_initializer())
    gamma = get_variable('gamma', (x.out_dim,), initializer=ones_initializer())

  def f(v, bt, gm):
    mean = v.mean(dim=-1, keepdim=True)
    var = ((v - mean) ** 2).mean(dim=-1, keepdim=True)
    inv = torch.rsqrt(var + 1e-12) * gm            # tf.nn.batch_normalization(x, mean, var, beta, gamma, 1e-12)
    return v * inv + (bt - mean * inv)
  out = Node(f, [x, beta, gamma], 'layer_norm')
  out.out_dim = x.out_dim
  return out


class _Layers(object):
  dense = staticmethod(_dense)


class _ContribLayers(object):
  layer_norm = staticmethod(_layer_norm)


class _Contrib(object):
  layers = _ContribLayers()


layers = _Layers()
contrib = _Contrib()


# ---------------------------------------------------------------------------------------------------------------------
# optimiser
# ---------------------------------------------------------------------------------------------------------------------
class _AdamStep(Node):
  def __init__(self, opt, loss, var_list):
    Node.__init__(self, None, [loss], 'adam_step')
    self.opt, self.var_list = opt, list(var_list)


class AdamOptimizer(object):
  """tf.train.AdamOptimizer(learning_rate) with TF's defaults beta1 = 0.9, beta2 = 0.999, epsilon = 1e-8."""

  def __init__(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
    self.lr, self.b1, self.b2, self.eps = float(learning_rate), float(beta1), float(beta2), float(epsilon)
    self.slots = {}              # Variable -> (m Variable, v Variable)
    self.b1p = self.b2p = None

  def minimize(self, loss, var_list=None):
    # slot and beta-power variables are global variables like any other: `<var>/Adam`, `<var>/Adam_1` live in the
    # variable's scope (so Model.vars sees them once they exist), the beta powers at the root
    def fresh(name, shape_, init):
      k, full = 0, name
      while full in _variables:
        k += 1
        full = '%s_%d' % (name, k)
      v = _variables[full] = Variable(full, shape_, init, False)
      return v
    for v in var_list:
      base = v.name[:-2]
      self.slots[v] = (fresh(base + '/Adam', v.shape_, zeros_initializer()), fresh(base + '/Adam_1', v.shape_, zeros_initializer()))
    self.b1p = fresh('beta1_power', (), lambda shape_, b=self.b1: np.float32(b))
    self.b2p = fresh('beta2_power', (), lambda shape_, b=self.b2: np.float32(b))
    return _AdamStep(self, loss, var_list)

  def variables(self):
    out = []
    for m, s in self.slots.values():
      out += [m, s]
    return out + [self.b1p, self.b2p]


class _Train(object):
  AdamOptimizer = AdamOptimizer


train = _Train()


# ---------------------------------------------------------------------------------------------------------------------
# session
# ---------------------------------------------------------------------------------------------------------------------
class Session(object):
  def run(self, fetches, feed_dict=None):
    single = not isinstance(fetches, (list, tuple))
    flist = [fetches] if single else list(fetches)
    feed = {k: torch.as_tensor(np.asarray(v, dtype=np.float32)) for k, v in (feed_dict or {}).items()}
    cache, writes, alive = {}, [], []           # `alive` keeps every leaf dict referenced: their ids key the cache

    def ev(node, leaves):
      """Value of `node`; `leaves` (Variable -> leaf tensor) substitutes variables when a gradient is being taken."""
      key = (id(node), id(leaves))
      if key in cache:
        return cache[key]
      if isinstance(node, Placeholder):
        if node not in feed:
          raise KeyError('placeholder %s was not fed' % node.name)
        val = feed[node]
      elif isinstance(node, Variable):
        if node.value is None:
          raise RuntimeError('variable %s is not initialised' % node.name)
        val = leaves[node] if (leaves is not None and node in leaves) else node.value
      elif isinstance(node, Assign):
        val = ev(node.inputs[0], leaves).detach().clone()
        writes.append((node.var, val))
      elif isinstance(node, Group):
        for op in node.inputs:
          ev(op, leaves)
        val = None
      elif isinstance(node, _AdamStep):
        val = adam(node)
      else:
        val = node.fn(*[ev(i, leaves) for i in node.inputs])
      cache[key] = val
      return val

    def adam(step):
      opt = step.opt
      leaves = {v: v.value.detach().clone().requires_grad_(True) for v in step.var_list}
      alive.append(leaves)
      loss = ev(step.inputs[0], leaves)
      grads = torch.autograd.grad(loss, [leaves[v] for v in step.var_list], allow_unused=True)
      b1p, b2p = float(opt.b1p.value), float(opt.b2p.value)
      lr_t = np.float32(opt.lr * np.sqrt(1.0 - b2p) / (1.0 - b1p))
      for v, g in zip(step.var_list, grads):
        g = torch.zeros_like(v.value) if g is None else g
        m_var, s_var = opt.slots[v]
        m = m_var.value * opt.b1 + g * (1.0 - opt.b1)
        s = s_var.value * opt.b2 + g * g * (1.0 - opt.b2)
        writes.append((m_var, m.detach()))
        writes.append((s_var, s.detach()))
        writes.append((v, (v.value - lr_t * m / (torch.sqrt(s) + opt.eps)).detach()))
      writes.append((opt.b1p, torch.as_tensor(np.float32(b1p * opt.b1))))
      writes.append((opt.b2p, torch.as_tensor(np.float32(b2p * opt.b2))))
      return None

    with torch.enable_grad():
      results = [ev(f, None) for f in flist]
    for var, val in writes:                          # every fetch saw the values from before this call
      var.value = val.to(torch.float32).reshape(var.shape_) if var.shape_ else val.to(torch.float32).reshape(())
    out = [None if r is None else r.detach().numpy().copy() for r in results]
    return out[0] if single else out


def global_variables():
  return list(_variables.values())


class _Summary(object):
  @staticmethod
  def scalar(*a, **kw):
    return None


summary = _Summary()
