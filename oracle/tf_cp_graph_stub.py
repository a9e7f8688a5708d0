"""TEST INFRASTRUCTURE: a NumPy graph with just enough of the TF-1.x Graph / Operation / Tensor / Session surface for the
reference's channel-pruner feature sampling (learners/channel_pruning/channel_pruner.py:215-227, 263-412, 579-586 and the
graph walks of model_wrapper.py:60-135, 195-254, 304-341) to be EXECUTED over it, and a small pre-activation residual
network + depthwise tail built on it (`build_standin`).  The same object also answers the oracle's plain interface (`ops()`,
`run(images, names)`), so the reference-executed fixture and oracle/cp_features_oracle.py see the identical network.

TF primitives restated here (float32 NumPy): Conv2D / DepthwiseConv2dNative (NHWC, SAME = total//2 before), FusedBatchNorm in
training mode (batch mean, biased variance), Relu / Relu6, Add, Pad, extract_image_patches (shared with the oracle)."""
import contextlib

import numpy as np

from oracle.cp_features_oracle import extract_image_patches


class Dim(object):
  def __init__(self, v):
    self.value = v


class Tensor(object):
  def __init__(self, graph, op, name, shape):
    self.graph, self.op, self.name = graph, op, name
    self.shape = [Dim(v) for v in shape]

  def consumers(self):
    return [o for o in self.graph.operations if any(t is self for t in o.inputs)]


class Operation(object):
  def __init__(self, graph, name, type_, inputs, fn, out_shape, attrs=None):
    self.graph, self.name, self.type, self.inputs, self.fn = graph, name, type_, list(inputs), fn
    self.attrs = dict(attrs or {})
    self.outputs = [Tensor(graph, self, name + ':0', out_shape)]
    graph.operations.append(self)

  def get_attr(self, key):
    v = self.attrs[key]
    return list(v) if isinstance(v, list) else v


class Graph(object):
  def __init__(self):
    self.operations = []

  @contextlib.contextmanager
  def as_default(self):
    yield self

  def get_operations(self):
    return list(self.operations)

  def get_tensor_by_name(self, name):
    for o in self.operations:
      if o.outputs[0].name == name:
        return o.outputs[0]
    raise KeyError(name)

  # -- the oracle's plain view --------------------------------------------------------------------------------------------------
  def ops(self):
    return [(o.name, o.type, [t.op.name for t in o.inputs]) for o in self.operations]


class Session(object):
  """run(fetches, feed_dict): fetches = tensor names / Tensors (or a list of them); a fetch that is a `Queue` pops its next
  element (the reference pulls its batches with sess.run([images, labels]))."""

  def __init__(self, graph):
    self.graph = graph

  def _eval(self, t, cache, feed):
    if t.name in cache:
      return cache[t.name]
    for k, v in feed.items():
      if k is t:
        cache[t.name] = np.asarray(v, np.float32)
        return cache[t.name]
    vals = [self._eval(i, cache, feed) for i in t.op.inputs]
    cache[t.name] = t.op.fn(*vals)
    return cache[t.name]

  def run(self, fetches, feed_dict=None):
    single = not isinstance(fetches, (list, tuple))
    cache, feed, out = {}, dict(feed_dict or {}), []
    for f in ([fetches] if single else fetches):
      if isinstance(f, Queue):
        out.append(f.pop())
        continue
      t = self.graph.get_tensor_by_name(f) if isinstance(f, str) else f
      out.append(self._eval(t, cache, feed))
    return out[0] if single else out


class Queue(object):
  def __init__(self, items):
    self.items, self.i = list(items), 0

  def pop(self):
    v = self.items[self.i % len(self.items)]
    self.i += 1
    return v


# -- op constructors ------------------------------------------------------------------------------------------------------------
def _same(size, k, s):
  out = -(-size // s)
  total = max((out - 1) * s + k - size, 0)
  return out, total // 2, total - total // 2


def _conv_fn(w, stride, padding, depthwise=False):
  kh, kw = w.shape[0], w.shape[1]

  def fn(x, _w=None):
    wcur = fn.weight
    p = extract_image_patches(x, kh, kw, stride, stride, padding)          # [B, Ho, Wo, kh*kw*C]
    if depthwise:
      C = x.shape[3]
      return np.einsum('bhwkc,kc->bhwc', p.reshape(p.shape[:3] + (kh * kw, C)), wcur.reshape(kh * kw, C)).astype(np.float32)
    return (p.reshape(-1, p.shape[-1]) @ wcur.reshape(-1, wcur.shape[-1])).reshape(p.shape[:3] + (wcur.shape[-1],)).astype(np.float32)
  fn.weight = w
  return fn


class Net(object):
  """Builder with TF-style creation order; `x` arguments and results are Tensors of the stand-in graph."""

  def __init__(self, rng, image_shape):
    self.g = Graph()
    self.rng = rng
    self.kernels = {}
    self.mem_images = Operation(self.g, 'mem_images', 'Placeholder', [], None, image_shape).outputs[0]

  def _shape(self, t):
    return [d.value for d in t.shape]

  def conv(self, x, name, k, cout, stride=1, padding='SAME'):
    B, H, W, C = self._shape(x)
    w = (self.rng.randn(k, k, C, cout) * (1.0 / np.sqrt(k * k * C))).astype(np.float32)
    read = Operation(self.g, name + '/kernel/read', 'Identity', [], lambda: None, [k, k, C, cout]).outputs[0]
    if padding == 'SAME':
      ho, wo = _same(H, k, stride)[0], _same(W, k, stride)[0]
    else:
      ho, wo = (H - k) // stride + 1, (W - k) // stride + 1
    fn = _conv_fn(w, stride, padding)
    op = Operation(self.g, name + '/Conv2D', 'Conv2D', [x, read], fn, [B, ho, wo, cout],
                   {'padding': padding.encode(), 'strides': [1, stride, stride, 1], 'data_format': b'NHWC'})
    self.kernels[op.name] = fn
    return op.outputs[0]

  def depthwise(self, x, name, k, stride=1):
    B, H, W, C = self._shape(x)
    w = (self.rng.randn(k, k, C, 1) * (1.0 / k)).astype(np.float32)
    read = Operation(self.g, name + '/depthwise_weights/read', 'Identity', [], lambda: None, [k, k, C, 1]).outputs[0]
    ho, wo = _same(H, k, stride)[0], _same(W, k, stride)[0]
    op = Operation(self.g, name + '/depthwise', 'DepthwiseConv2dNative', [x, read], _conv_fn(w, stride, 'SAME', True), [B, ho, wo, C])
    return op.outputs[0]

  def bn(self, x, name):
    C = self._shape(x)[3]
    gamma = (1.0 + 0.1 * self.rng.randn(C)).astype(np.float32)
    beta = (0.1 * self.rng.randn(C)).astype(np.float32)

    def fn(v):
      m = v.mean(axis=(0, 1, 2), dtype=np.float32)
      var = ((v - m) ** 2).mean(axis=(0, 1, 2), dtype=np.float32)
      return ((v - m) / np.sqrt(var + np.float32(1e-5)) * gamma + beta).astype(np.float32)
    return Operation(self.g, name + '/FusedBatchNorm', 'FusedBatchNorm', [x], fn, self._shape(x)).outputs[0]

  def relu(self, x, name, six=False):
    fn = (lambda v: np.clip(v, 0, 6)) if six else (lambda v: np.maximum(v, 0))
    return Operation(self.g, name, 'Relu6' if six else 'Relu', [x], fn, self._shape(x)).outputs[0]

  def add(self, a, b, name):
    return Operation(self.g, name, 'Add', [a, b], lambda u, v: u + v, self._shape(a)).outputs[0]

  def pad(self, x, name, before, after):
    B, H, W, C = self._shape(x)
    fn = lambda v: np.pad(v, ((0, 0), (before, after), (before, after), (0, 0)))
    return Operation(self.g, name, 'Pad', [x], fn, [B, H + before + after, W + before + after, C]).outputs[0]

  def fixed_padding_conv(self, x, name, k, cout, stride):
    """conv2d_fixed_padding (utils/external/resnet_model.py:92-103): explicit Pad + VALID when strided."""
    if stride > 1:
      total = k - 1
      x = self.pad(x, name + '/Pad', total // 2, total - total // 2)
    return self.conv(x, name, k, cout, stride, 'SAME' if stride == 1 else 'VALID')


def build_standin(seed=5, batch=3, hw=12):
  """Pre-activation bottleneck blocks (projection + stride 2, then identity) followed by a depthwise-separable tail; op
  creation order as in resnet_model.py:257-314 (projection shortcut first)."""
  net = Net(np.random.RandomState(seed), [batch, hw, hw, 3])
  x = net.fixed_padding_conv(net.mem_images, 'conv0', 3, 8, 1)
  # block 1: projection shortcut, stride 2
  pre = net.relu(net.bn(x, 'b1/bn1'), 'b1/Relu')
  shortcut = net.fixed_padding_conv(pre, 'b1/proj', 1, 16, 2)
  y = net.fixed_padding_conv(pre, 'b1/conv1', 1, 4, 1)
  y = net.relu(net.bn(y, 'b1/bn2'), 'b1/Relu_1')
  y = net.fixed_padding_conv(y, 'b1/conv2', 3, 4, 2)
  y = net.relu(net.bn(y, 'b1/bn3'), 'b1/Relu_2')
  y = net.fixed_padding_conv(y, 'b1/conv3', 1, 16, 1)
  x = net.add(y, shortcut, 'b1/add')
  # block 2: identity shortcut
  pre = net.relu(net.bn(x, 'b2/bn1'), 'b2/Relu')
  y = net.fixed_padding_conv(pre, 'b2/conv1', 1, 4, 1)
  y = net.relu(net.bn(y, 'b2/bn2'), 'b2/Relu_1')
  y = net.fixed_padding_conv(y, 'b2/conv2', 3, 4, 1)
  y = net.relu(net.bn(y, 'b2/bn3'), 'b2/Relu_2')
  y = net.fixed_padding_conv(y, 'b2/conv3', 1, 16, 1)
  x = net.add(y, x, 'b2/add')
  # depthwise-separable tail: the walk from 'tail/pw0' passes BN, Relu6 and the depthwise convolution and ends at a Conv2D
  y = net.conv(x, 'tail/pw0', 1, 12)
  y = net.relu(net.bn(y, 'tail/bn0'), 'tail/Relu6', six=True)
  y = net.depthwise(y, 'tail/dw', 3, 1)
  y = net.relu(net.bn(y, 'tail/bn1'), 'tail/Relu6_1', six=True)
  net.conv(y, 'tail/pw1', 1, 10)
  return net
