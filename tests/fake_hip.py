"""float32 torch / NumPy emulation of the HIP entry points (include/pocketflow_hip.h) for CPU-side tests.

`FakeHip` replaces the `hip` module object inside pocketflow_amd.graph / plan / losses / optim / learners
(monkeypatch), so that the HOST logic around the kernels -- autograd plumbing, launch plans, learner loops,
schedules, mask refreshes -- can be executed and checked exactly on a machine without a GPU.  The emulations
follow the documented semantics of each entry point; the segment (all-weights) kernels, the losses and the
optimisers delegate to the CPU oracle (oracle/pf_oracle.py), which is allowed here because this file is
test infrastructure.  Nothing under pocketflow_amd/ imports it.
"""
import numpy as np
import torch

from oracle import pf_oracle as O


def _rows(t, C):
  """[rows][C] view of a logical-NCHW / physical-NHWC tensor (or a flat [rows][C] one)."""
  if t.dim() == 4:
    return t.permute(0, 2, 3, 1).reshape(-1, C)
  return t.reshape(-1, C)


def _act(y, act):
  if act in ('Relu', 'relu'):
    return torch.relu(y)
  if act in ('Relu6', 'relu6'):
    return torch.clamp(y, 0, 6)
  return y


def _mask(u, act):
  if act in ('Relu', 'relu'):
    return (u > 0).float()
  if act in ('Relu6', 'relu6'):
    return ((u > 0) & (u < 6)).float()
  return torch.ones_like(u)


class FakeHip(object):
  """float32 torch emulation of the entry points graph.py calls; min/max slots hold two float32 bit patterns."""

  def __init__(self):
    self.calls = {}

  def _n(self, name):
    self.calls[name] = self.calls.get(name, 0) + 1

  # -- slots ----------------------------------------------------------------------------------------
  def minmax_slots_init(self, slots):
    slots.view(-1).copy_(torch.tensor([float('inf'), float('-inf')] * (slots.numel() // 2)).view(torch.int32))

  @staticmethod
  def _slot_get(slot):
    v = slot.view(torch.float32)
    return float(v[0]), float(v[1])

  @staticmethod
  def _quant(y, slot, bits):
    mn, mx = FakeHip._slot_get(slot)
    alpha, beta = (mx - mn) + 1e-10, mn
    k = float(2 ** bits - 1)
    return alpha * (torch.round((y - beta) / alpha * k) / k) + beta

  # -- BN forward --------------------------------------------------------------------------------------
  def bn_stats(self, x, rows, C, partial, nblk):
    self._n('bn_stats')
    xr = _rows(x, C).float()
    p = partial[:nblk * 4 * C].view(nblk, 4, C)
    p[:, 0:2] = 0
    p[:, 2] = float('inf')
    p[:, 3] = float('-inf')
    d = xr - xr[0:1]
    p[0, 0], p[0, 1], p[0, 2], p[0, 3] = d.sum(0), (d * d).sum(0), xr.min(0).values, xr.max(0).values

  def bn_finalize(self, partial, nblk, rows, C, piv, gamma, beta, mm, mv, momentum, eps, training, act, ss, mi, slot):
    self._n('bn_finalize')
    p = partial.reshape(-1)[:nblk * 4 * C].view(nblk, 4, C).double()
    pivot = _rows(piv, C)[0].double() if piv.numel() >= C and piv.dim() != 1 else piv.reshape(-1)[:C].double()
    s, q = p[:, 0].sum(0), p[:, 1].sum(0)
    mn, mx = p[:, 2].min(0).values.float(), p[:, 3].max(0).values.float()
    if training:
      m1 = s / rows
      var = (q / rows - m1 * m1).clamp_min(0)
      mean = (pivot + m1).float()
      unbiased = (var * (rows / max(rows - 1, 1))).float()
      mm.sub_((mm - mean) * (1 - momentum))
      mv.sub_((mv - unbiased) * (1 - momentum))
      var = var.float()
    else:
      mean, var = mm.clone(), mv.clone()
    invstd = 1.0 / torch.sqrt(var + eps)
    sc = gamma.detach() * invstd
    ss[0], ss[1] = sc, beta.detach() - mean * sc
    mi[0], mi[1] = mean, invstd
    if slot is not None:
      a, b = sc * mn + ss[1], sc * mx + ss[1]
      ymin, ymax = _act(torch.minimum(a, b), act).min(), _act(torch.maximum(a, b), act).max()
      cur = slot.view(torch.float32)
      cur[0], cur[1] = min(float(cur[0]), float(ymin)), max(float(cur[1]), float(ymax))

  def bn_eval_scale_shift(self, gamma, beta, mm, mv, eps, ss):
    sc = gamma.detach() / torch.sqrt(mv + eps)
    ss[0], ss[1] = sc, beta.detach() - mm * sc

  def _q_of(self, xr, ss, act, slot, bits, quantize):
    y = _act(xr * ss[0] + ss[1], act)
    return self._quant(y, slot, bits) if quantize else y

  def bn_act_quant_apply(self, x, q, rows, C, ss, act, slot, bits, quantize):
    self._n('bn_apply')
    _rows(q, C).copy_(self._q_of(_rows(x, C).float(), ss, act, slot, bits, quantize))

  # -- BN backward -------------------------------------------------------------------------------------------
  def bn_bwd_stats(self, dq, x, rows, C, ss, mi, act, partial, nblk):
    self._n('bn_bwd_stats')
    xr, g = _rows(x, C).float(), _rows(dq, C).float()
    dy = g * _mask(xr * ss[0] + ss[1], act)
    p = partial[:nblk * 2 * C].view(nblk, 2, C)
    p.zero_()
    p[0, 0], p[0, 1] = dy.sum(0), (dy * (xr - mi[0]) * mi[1]).sum(0)

  def bn_bwd_finalize(self, partial, nblk, C, dgamma, dbeta):
    p = partial.reshape(-1)[:nblk * 2 * C].view(nblk, 2, C)
    dbeta.copy_(p[:, 0].sum(0))
    dgamma.copy_(p[:, 1].sum(0))

  def bn_bwd_apply(self, dq, x, dx, rows, C, ss, mi, dgamma, dbeta, act, addend=None):
    self._n('bn_bwd_apply_add' if addend is not None else 'bn_bwd_apply')
    xr, g = _rows(x, C).float(), _rows(dq, C).float()
    dy = g * _mask(xr * ss[0] + ss[1], act)
    out = ss[0] * (dy - dbeta / rows - (xr - mi[0]) * mi[1] * dgamma / rows)
    if addend is not None:
      out = out + _rows(addend, C).float()
    _rows(dx, C).copy_(out)

  # -- fused 1x1 convolutions -------------------------------------------------------------------------------------
  def conv1x1_stats_groups(self, M, N, K=0, prologue=False):
    return 3

  def conv1x1_wrw_splits(self, M, N, K):
    return 2

  # -- few-channel RxS convolutions: the gather and its inverse (pf_im2col.hip) ------------------------------------------------
  @staticmethod
  def _padded_extent(size, n_out, k, stride, pad):
    return max(size + pad, (n_out - 1) * stride + k)

  def im2col(self, X, Xcol, B, H, Wd, C, R, S, stride, ph, pw, Ho, Wo):
    self._n('im2col')
    import torch.nn.functional as F
    x = X.detach().permute(0, 2, 3, 1).float()                      # [B][H][W][C]
    Hp, Wp = self._padded_extent(H, Ho, R, stride, ph), self._padded_extent(Wd, Wo, S, stride, pw)
    xp = F.pad(x, (0, 0, pw, Wp - pw - Wd, ph, Hp - ph - H))
    cols = [xp[:, r:r + (Ho - 1) * stride + 1:stride, s:s + (Wo - 1) * stride + 1:stride, :] for r in range(R) for s in range(S)]
    Xcol.copy_(torch.cat(cols, dim=3).reshape(B * Ho * Wo, R * S * C).to(Xcol.dtype))

  def col2im(self, dXcol, dX, B, H, Wd, C, R, S, stride, ph, pw, Ho, Wo):
    self._n('col2im')
    d = dXcol.detach().float().view(B, Ho, Wo, R * S, C)
    Hp, Wp = self._padded_extent(H, Ho, R, stride, ph), self._padded_extent(Wd, Wo, S, stride, pw)
    dxp = torch.zeros((B, Hp, Wp, C), dtype=torch.float32)
    for r in range(R):
      for s in range(S):
        dxp[:, r:r + (Ho - 1) * stride + 1:stride, s:s + (Wo - 1) * stride + 1:stride, :] += d[:, :, :, r * S + s, :]
    dX.copy_(dxp[:, ph:ph + H, pw:pw + Wd, :].permute(0, 3, 1, 2).to(dX.dtype))

  @staticmethod
  def _gather(x, K, geom):
    if geom is None:
      return _rows(x, K).float()
    ho, wo, h, w, s = geom
    return x.permute(0, 2, 3, 1)[:, ::s, ::s, :][:, :ho, :wo, :].reshape(-1, K).float()

  def conv1x1_fwd(self, X, W, Y, M, N, K, R=None, scale_shift=None, act=None, slot=None, bits=8, partial=None,
                  geom=None, ymap=False, out_scale_shift=None, out_act=None):
    self._n('conv1x1_fwd' if scale_shift is not None else 'conv1x1_plain')
    if out_scale_shift is not None:
      self._n('conv_out_affine')
    if ymap:                                   # backward-data of a strided conv: scatter rows
      ho, wo, h, w, s = geom
      out = _rows(X, K).float() @ W.float().t()
      Y.permute(0, 2, 3, 1)[:, ::s, ::s, :][:, :ho, :wo, :] = out.view(Y.shape[0], ho, wo, N)
      return
    xr = self._gather(X, K, geom)
    if scale_shift is not None:
      xr = self._q_of(xr, scale_shift, act, slot, bits, slot is not None)
    y = xr @ W.float().t()
    if R is not None:
      y = y + _rows(R, N).float()
    if out_scale_shift is not None:              # the consumer's inference-mode BN + activation (pf_conv1x1_fwd_affine)
      y = self._q_of(y, out_scale_shift, out_act, None, 8, False)
    _rows(Y, N).copy_(y)
    if partial is not None:
      partial[:, 0:2] = 0
      partial[:, 2] = float('inf')
      partial[:, 3] = float('-inf')
      partial[1, 0], partial[1, 1], partial[1, 2], partial[1, 3] = y.sum(0), (y * y).sum(0), y.min(0).values, y.max(0).values

  def conv1x1_bwd_data_bnstats(self, dY, Wt, dQ, bn_x, bn_ss, bn_mi, bn_act, partial, M, N, K):
    self._n('conv1x1_bwd_data_bnstats')
    dq = _rows(dY, N).float() @ Wt.float().t()
    _rows(dQ, K).copy_(dq)
    xr = _rows(bn_x, K).float()
    dy = dq * _mask(xr * bn_ss[0] + bn_ss[1], bn_act)
    partial.zero_()
    partial[2, 0], partial[2, 1] = dy.sum(0), (dy * (xr - bn_mi[0]) * bn_mi[1]).sum(0)

  def conv1x1_wrw(self, dY, X, dW, workspace, M, N, K, scale_shift=None, act=None, slot=None, bits=8, geom=None):
    self._n('conv1x1_wrw')
    xr = self._gather(X, K, geom)
    if scale_shift is not None:
      xr = self._q_of(xr, scale_shift, act, slot, bits, slot is not None)
    dW.copy_(_rows(dY, N).float().t() @ xr)

  def seg_transpose(self, src_flat, dst_flat, tiles, n_tiles):
    import numpy as np
    from pocketflow_amd import hip as real
    t = np.frombuffer(tiles.numpy().tobytes(), dtype=real.TILE_DTYPE)
    for r in t:
      o1, i1 = min(int(r['o0']) + 64, int(r['O'])), min(int(r['i0']) + 64, int(r['I']))
      for o in range(int(r['o0']), o1):
        src = src_flat[int(r['src_off']) + o * int(r['src_ld']) + int(r['i0']):int(r['src_off']) + o * int(r['src_ld']) + i1]
        dst_flat[int(r['dst_off']) + int(r['i0']) * int(r['dst_ld']) + o:int(r['dst_off']) + (i1 - 1) * int(r['dst_ld']) + o + 1:int(r['dst_ld'])] = src

  def conv2d_wrw_splits(self, M, N, C, taps):
    return 2

  def conv2d_wrw(self, dY, X, dW, workspace, imgs, H, Wd, C, N, th, tw, stride, pad_h, pad_w, Ho, Wo):
    """dW: KRSC memory view [N][th][tw][C]."""
    self._n('conv2d_wrw')
    import torch
    w0 = torch.zeros((N, C, th, tw), dtype=torch.float32)
    g = torch.ops.aten.convolution_backward(dY.float(), X.float(), w0, None, [stride, stride], [pad_h, pad_w], [1, 1],
                                            False, [0, 0], 1, [False, True, False])[1]
    dW.copy_(g.permute(0, 2, 3, 1))

  # -- RxS convolutions (pf_igemm.hip) ------------------------------------------------------------------------------------
  def conv2d_stats_groups(self, M, N, geom=None):
    return 3

  def conv2d_fwd(self, X, W, Y, imgs, H, Wd, C, N, th, tw, stride, pad_h, pad_w, Ho, Wo, R=None, partial=None, bn_x=None,
                 bn_scale_shift=None, bn_mean_invstd=None, bn_act=None, out_scale_shift=None, out_act=None):
    """X: logical NCHW over NHWC memory, W: [N][th][tw][C]; same float32 arithmetic as a dense convolution."""
    self._n('conv2d_fwd' if bn_x is None else 'conv2d_bwd_data_bnstats')
    import torch.nn.functional as F
    y = F.conv2d(X.float(), W.float().permute(0, 3, 1, 2), stride=stride, padding=(pad_h, pad_w))
    yr = y.permute(0, 2, 3, 1).reshape(-1, N)
    if R is not None:
      yr = yr + _rows(R, N).float()
    if out_scale_shift is not None:
      self._n('conv_out_affine')
      yr = self._q_of(yr, out_scale_shift, out_act, None, 8, False)
    _rows(Y, N).copy_(yr)
    if partial is None:
      return
    if bn_x is not None:
      xr = _rows(bn_x, N).float()
      dy = yr * _mask(xr * bn_scale_shift[0] + bn_scale_shift[1], bn_act)
      partial.zero_()
      partial[2, 0], partial[2, 1] = dy.sum(0), (dy * (xr - bn_mean_invstd[0]) * bn_mean_invstd[1]).sum(0)
    else:
      partial[:, 0:2] = 0
      partial[:, 2] = float('inf')
      partial[:, 3] = float('-inf')
      partial[1, 0], partial[1, 1], partial[1, 2], partial[1, 3] = yr.sum(0), (yr * yr).sum(0), yr.min(0).values, yr.max(0).values

  # -- depthwise 3x3 (pf_depthwise.hip): the tensors arrive as logical NCHW / physical NHWC torch views ------------------------
  def depthwise_supported(self, C, k, stride):
    return k == 3 and C % 8 == 0 and C >= 8 and 256 % (C // 8) == 0 and stride in (1, 2)

  def depthwise_groups(self, B, Ho, Wo, C):
    return 2

  @staticmethod
  def _dw_pad(x, H, Wd, k, stride, ph, pw, Ho, Wo):
    import torch.nn.functional as F
    pb_h, pb_w = max((Ho - 1) * stride + k - H - ph, 0), max((Wo - 1) * stride + k - Wd - pw, 0)
    return F.pad(x, (pw, pb_w, ph, pb_h))

  def depthwise_fwd(self, X, W, Y, B, H, Wd, C, k, stride, pad_h, pad_w, Ho, Wo, partial=None):
    self._n('depthwise_fwd')
    import torch.nn.functional as F
    y = F.conv2d(self._dw_pad(X.float(), H, Wd, k, stride, pad_h, pad_w, Ho, Wo), W.float().reshape(C, 1, k, k), stride=stride, groups=C)
    Y.copy_(y.to(Y.dtype))
    if partial is not None:
      yr = Y.float().permute(0, 2, 3, 1).reshape(-1, C)
      partial[:, 0:2] = 0
      partial[:, 2] = float('inf')
      partial[:, 3] = float('-inf')
      partial[1, 0], partial[1, 1], partial[1, 2], partial[1, 3] = yr.sum(0), (yr * yr).sum(0), yr.min(0).values, yr.max(0).values

  def depthwise_bwd_data(self, dY, W, dX, B, H, Wd, C, k, stride, pad_h, pad_w, Ho, Wo):
    self._n('depthwise_bwd_data')
    import torch.nn.functional as F
    with torch.enable_grad():                    # (called from inside a backward pass, where grad mode is off)
      x = torch.zeros(B, C, H, Wd, requires_grad=True)
      y = F.conv2d(self._dw_pad(x, H, Wd, k, stride, pad_h, pad_w, Ho, Wo), W.detach().float().reshape(C, 1, k, k), stride=stride, groups=C)
      (g,) = torch.autograd.grad(y, x, dY.float())
    dX.copy_(g.to(dX.dtype))

  def depthwise_wrw(self, dY, X, dW, slabs, B, H, Wd, C, k, stride, pad_h, pad_w, Ho, Wo):
    self._n('depthwise_wrw')
    import torch.nn.functional as F
    with torch.enable_grad():
      w = torch.zeros(C, 1, k, k, requires_grad=True)
      y = F.conv2d(self._dw_pad(X.detach().float(), H, Wd, k, stride, pad_h, pad_w, Ho, Wo), w, stride=stride, groups=C)
      (g,) = torch.autograd.grad(y, w, dY.float())
    dW.copy_(g.reshape(dW.shape).to(dW.dtype))

  # -- the ResNet stem (pf_stem.hip) --------------------------------------------------------------------------------------
  def conv_stem_supported(self, H, Wd, C, N, k, stride, pad):
    return C == 3 and N == 64 and k == 7 and stride == 2 and pad == 3 and H % 2 == 0 and Wd % 32 == 0 and 32 <= Wd <= 1024

  def conv_stem_wrw_slabs(self, imgs, H, Wd):
    return 2 if self.conv_stem_supported(H, Wd, 3, 64, 7, 2, 3) and Wd <= 256 else 0

  def conv_stem_wrw(self, dY, X, dW, workspace, imgs, H, Wd):
    self._n('conv_stem_wrw')
    w0 = torch.zeros(64, 3, 7, 7)
    g = torch.ops.aten.convolution_backward(dY.float(), X.float(), w0, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                            [False, True, False])[1]
    dW.copy_(g.permute(0, 2, 3, 1))

  def conv_stem_fwd(self, X, W, Y, imgs, H, Wd):
    self._n('conv_stem_fwd')
    import torch.nn.functional as F
    y = F.conv2d(X.float(), W.float().permute(0, 3, 1, 2), stride=2, padding=3)
    _rows(Y, 64).copy_(y.permute(0, 2, 3, 1).reshape(-1, 64))

  # -- the MobileNet stem (pf_stem3.hip): 3x3 / stride 2 with TensorFlow 'SAME' padding given by its FRONT pads -------------
  def conv_stem3_supported(self, H, Wd, C, N, k, stride, pad_h, pad_w, Ho, Wo):
    return (C == 3 and N in (16, 32) and k == 3 and stride == 2 and H > 0 and 32 <= Wd <= 1024 and Wd % 2 == 0 and pad_h in (0, 1)
            and pad_w in (0, 1) and Wo > 0 and Wo % 16 == 0 and Ho > 0 and 2 * Wo <= Wd + 2 and 2 * (Ho - 1) - pad_h < H)

  def conv_stem3_wrw_slabs(self, imgs, H, Wd, N, pad_h, pad_w, Ho, Wo):
    return 2 if self.conv_stem3_supported(H, Wd, 3, N, 3, 2, pad_h, pad_w, Ho, Wo) else 0

  @staticmethod
  def _stem3_padded(X, H, Wd, pad_h, pad_w, Ho, Wo):
    import torch.nn.functional as F
    back_h = max((Ho - 1) * 2 + 3 - H - pad_h, 0)
    back_w = max((Wo - 1) * 2 + 3 - Wd - pad_w, 0)
    return F.pad(X.float(), (pad_w, back_w, pad_h, back_h))

  def conv_stem3_fwd(self, X, W, Y, imgs, H, Wd, N, pad_h, pad_w, Ho, Wo):
    self._n('conv_stem3_fwd')
    import torch.nn.functional as F
    y = F.conv2d(self._stem3_padded(X, H, Wd, pad_h, pad_w, Ho, Wo), W.float().permute(0, 3, 1, 2), stride=2)[:, :, :Ho, :Wo]
    _rows(Y, N).copy_(y.permute(0, 2, 3, 1).reshape(-1, N))

  def conv_stem3_wrw(self, dY, X, dW, workspace, imgs, H, Wd, N, pad_h, pad_w, Ho, Wo):
    self._n('conv_stem3_wrw')
    xp = self._stem3_padded(X, H, Wd, pad_h, pad_w, Ho, Wo)
    w0 = torch.zeros(N, 3, 3, 3)
    g = torch.ops.aten.convolution_backward(dY.float(), xp, w0, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1,
                                            [False, True, False])[1]
    dW.copy_(g.permute(0, 2, 3, 1))



# =================================================================================================
# learner-level entry points (segment kernels, losses, optimisers, sparsification) on top of FakeHip
# =================================================================================================

def _real_constants():
  from pocketflow_amd import hip as real
  return {k: getattr(real, k) for k in ('PF_F32', 'PF_BF16', 'PF_ACT_NONE', 'PF_ACT_RELU', 'PF_ACT_RELU6',
                                        'PF_BUCKET_TENSOR', 'PF_BUCKET_CHANNEL', 'PF_BUCKET_SPLIT', 'PF_CHUNK',
                                        'SEG_DTYPE', 'BLOCK_DTYPE', 'ACT_CODES', 'IMAGE_DESC_DTYPE')}


class FakeHipFull(FakeHip):
  def __init__(self):
    super(FakeHipFull, self).__init__()
    for k, v in _real_constants().items():
      setattr(self, k, v)
    self._nuq_info = {}

  # -- stand-alone activations (LeNet, ResNet-v1) ---------------------------------------------------------
  def minmax_tensor(self, x, slot, act=None):
    t = _act(x.float(), act)
    cur = slot.view(torch.float32)
    cur[0], cur[1] = min(float(cur[0]), float(t.min())), max(float(cur[1]), float(t.max()))

  def minmax_decode(self, slots):
    v = slots.view(torch.float32).view(-1, 2)
    return torch.stack([(v[:, 1] - v[:, 0]) + 1e-10, v[:, 0]], dim=1)

  def uq_apply(self, x, y, slot, bits, act=None):
    y.copy_(self._quant(_act(x.float(), act), slot, bits).to(y.dtype))

  def act_grad(self, g, u, dx, act):
    dx.copy_(g * _mask(u.float(), act))

  # -- segment (all-weights) kernels: delegate to the oracle, tensor by tensor ------------------------------------
  def _segs(self, segs):
    return np.frombuffer(segs.cpu().numpy().tobytes(), dtype=self.SEG_DTYPE)

  @staticmethod
  def _to_hwio(flat, sg):
    RS, I, Oc = int(sg['RS']), int(sg['I']), int(sg['O'])
    if int(sg['layout']) == 1:                       # depthwise: storage [C][RS] -> [RS][C][1]
      return flat.reshape(I, RS).T.reshape(RS, I, 1)
    return flat.reshape(Oc, RS, I).transpose(1, 2, 0)            # KRSC [O][RS][I] -> [RS][I][O]

  @staticmethod
  def _to_storage(hwio, sg):
    RS, I, Oc = int(sg['RS']), int(sg['I']), int(sg['O'])
    if int(sg['layout']) == 1:
      return np.ascontiguousarray(hwio.reshape(RS, I).T).reshape(-1)
    return np.ascontiguousarray(hwio.reshape(RS, I, Oc).transpose(2, 0, 1)).reshape(-1)

  def _bucket_args(self, sg):
    mode = int(sg['mode'])
    if mode == self.PF_BUCKET_TENSOR:
      return False, 'channel', 0
    if mode == self.PF_BUCKET_CHANNEL:
      return True, 'channel', 0
    return True, 'split', int(sg['bucket_size'])

  def seg_minmax(self, w_flat, segs, blocks, n_blocks, slots):
    # the apply emulations recompute their ranges; the slots are filled for what READS them (plan.alpha_beta(): logging,
    # the integer export) in this emulation's own encoding: float32 (min, max) pairs, see minmax_decode above
    w = w_flat.detach().float().numpy()
    v = slots.view(torch.float32).view(-1, 2)
    for sg in self._segs(segs):
      off, n = int(sg['offset']), int(sg['len'])
      ub, bt, bs = self._bucket_args(sg)
      x = self._to_hwio(w[off:off + n], sg)
      if ub and bt == 'split':
        x = O.split_bucket(x, bs)[0]
      elif ub:
        x = O.channel_bucket(x)[0]
      else:
        x = x.reshape(-1, 1)
      so = int(sg['slot_offset'])
      v[so:so + x.shape[1], 0] = torch.from_numpy(np.ascontiguousarray(x.min(axis=0)))
      v[so:so + x.shape[1], 1] = torch.from_numpy(np.ascontiguousarray(x.max(axis=0)))

  def seg_uq_apply(self, w_flat, qw_flat, segs, blocks, n_blocks, slots):
    w = w_flat.detach().numpy()
    out = qw_flat.detach().numpy() if qw_flat.dtype == torch.float32 else None
    for sg in self._segs(segs):
      off, n, bits = int(sg['offset']), int(sg['len']), int(sg['bits'])
      src = w[off:off + n]
      if bits > 0:
        ub, bt, bs = self._bucket_args(sg)
        q, _ = O.uniform_quantize(self._to_hwio(src, sg), bits, 'weight', ub, bt, bs)
        res = self._to_storage(q, sg)
      else:
        res = src
      if out is not None:
        out[off:off + n] = res
      else:
        qw_flat[off:off + n] = torch.from_numpy(np.ascontiguousarray(res)).to(qw_flat.dtype)

  def _codebook(self, codebooks, sg, ub):
    k, nb, cb = 2 ** int(sg['bits']), int(sg['n_bucket']), int(sg['cb_offset'])
    c = codebooks.detach().numpy()[cb:cb + k * nb]
    return c.reshape(k, nb) if ub else c.reshape(k)

  def seg_nuq_apply(self, w_flat, qw_flat, idx_flat, codebooks, segs, blocks, n_blocks, slots):
    w = w_flat.detach().numpy()
    self._nuq_info = {}
    for s, sg in enumerate(self._segs(segs)):
      off, n, bits = int(sg['offset']), int(sg['len']), int(sg['bits'])
      src = w[off:off + n]
      if bits > 0:
        ub, bt, bs = self._bucket_args(sg)
        q, info = O.nuq_quantize(self._to_hwio(src, sg), bits, self._codebook(codebooks, sg, ub), ub, bt, bs)
        self._nuq_info[s] = (info, ub, bt, bs)
        res = self._to_storage(q, sg)
        if not ub and idx_flat is not None:              # per-tensor codebooks: the codeword index of every element, storage order
          idx_flat[off:off + n] = torch.from_numpy(np.ascontiguousarray(self._to_storage(info['idx'].astype(np.float32), sg))).to(idx_flat.dtype)
      else:
        res = src
      qw_flat[off:off + n] = torch.from_numpy(np.ascontiguousarray(res)).to(qw_flat.dtype)

  def seg_nuq_codebook_grad(self, g_flat, idx_flat, dcodebooks, acc_ws, segs, blocks, n_blocks, slots):
    g = g_flat.detach().float().numpy()
    dc_all = dcodebooks.detach().numpy()
    for s, sg in enumerate(self._segs(segs)):
      if s not in self._nuq_info:
        continue
      info, ub, bt, bs = self._nuq_info[s]
      off, n = int(sg['offset']), int(sg['len'])
      _, dc = O.nuq_backward(self._to_hwio(g[off:off + n], sg), info, ub, bt, bs)
      k, nb, cb = 2 ** int(sg['bits']), int(sg['n_bucket']), int(sg['cb_offset'])
      dc_all[cb:cb + k * nb] += np.asarray(dc, np.float32).reshape(-1)

  def seg_normalize(self, w_flat, xn_out, segs, seg_index, slots):
    sg = self._segs(segs)[seg_index]
    off, n = int(sg['offset']), int(sg['len'])
    hw = self._to_hwio(w_flat.detach().numpy()[off:off + n], sg)
    ub, bt, bs = self._bucket_args(sg)
    if not ub:
      xn, _, _ = O.scale(hw, None)
    elif bt == 'channel':
      xb, _, _ = O.channel_bucket(hw)
      xn, _, _ = O.scale(xb, 0)
      xn = xn.reshape(hw.shape)
    else:
      xb, _, pad = O.split_bucket(hw, bs)
      xn, _, _ = O.scale(xb, 0)
      xn = xn.reshape(-1)
      xn = (xn[:-pad] if pad else xn).reshape(hw.shape)
    xn_out.copy_(torch.from_numpy(self._to_storage(np.asarray(xn, np.float32), sg)))

  # -- losses ----------------------------------------------------------------------------------------------------------
  def ce_distill_fwd_bwd(self, z_s, labels, z_t, tempr, loss_w, losses, dz_s, row_ws):
    zs = z_s.detach().float().numpy()
    ce, dz = O.softmax_cross_entropy(labels.detach().numpy(), zs)
    dl = np.float32(0)
    if z_t is not None:
      dl, ddz = O.distill_loss(zs, z_t.detach().float().numpy(), tempr, loss_w)
      dz = dz + ddz
    losses[0], losses[1] = float(ce), float(dl)
    dz_s.copy_(torch.from_numpy(np.asarray(dz, np.float32)).to(dz_s.dtype))

  # -- optimisers ----------------------------------------------------------------------------------------------------------
  @staticmethod
  def _eff_grad(p, g, mask, n_decay, wd, g_scale):
    ge = g.detach().float().numpy() * np.float32(g_scale)
    pn = p.detach().numpy()
    ge[:n_decay] = ge[:n_decay] + np.float32(wd) * pn[:n_decay]
    if mask is not None:
      ge = ge * mask.detach().numpy()
    return ge.astype(np.float32), pn

  def adam_flat(self, p, g, m, v, mask, n_decay, wd, g_scale, lr, beta1, beta2, eps, beta1_power, beta2_power):
    ge, pn = self._eff_grad(p, g, mask, n_decay, wd, g_scale)
    one = np.float32(1)
    alpha_t = np.float32(np.float32(lr) * np.sqrt(one - np.float32(beta2_power)) / (one - np.float32(beta1_power)))
    mn, vn = m.numpy(), v.numpy()
    mn += (ge - mn) * (one - np.float32(beta1))
    vn += (ge * ge - vn) * (one - np.float32(beta2))
    pn -= (mn * alpha_t) / (np.sqrt(vn) + np.float32(eps))

  # -- chn-pruned-gpu: the proximal step ------------------------------------------------------------------------------------
  def prox_groups(self, rows, I):
    return 1

  def prox_norms(self, w, g, lr, rows, I, partial, norms):
    new = w.detach().float().view(rows, I) - np.float32(lr) * g.detach().float().view(rows, I)
    norms[:I] = torch.sqrt((new * new).sum(dim=0))

  def prox_apply(self, w, g, lr, rows, I, norms, thr):
    new = w.detach().view(rows, I) - np.float32(lr) * g.detach().float().view(rows, I)
    shrk = torch.clamp(1.0 - thr[0] / norms[:I], min=0.0)
    shrk = torch.where(torch.isnan(shrk), torch.zeros_like(shrk), shrk)
    w.view(rows, I).copy_(new * shrk)

  def set_floats(self, dst, a, b=0.0, c=0.0, d=0.0):
    dst[:4] = torch.tensor([a, b, c, d], dtype=torch.float32)

  def adam_flat_dev(self, p, g, m, v, mask, n_decay, wd, g_scale, hp, beta1, beta2, eps):
    ge, pn = self._eff_grad(p, g, mask, n_decay, wd, g_scale)
    one = np.float32(1)
    alpha_t = np.float32(hp[0].item())
    mn, vn = m.numpy(), v.numpy()
    mn += (ge - mn) * (one - np.float32(beta1))
    vn += (ge * ge - vn) * (one - np.float32(beta2))
    pn -= (mn * alpha_t) / (np.sqrt(vn) + np.float32(eps))

  def momentum_flat_dev(self, p, g, acc, mask, n_decay, wd, g_scale, hp, momentum):
    self.momentum_flat(p, g, acc, mask, n_decay, wd, g_scale, float(hp[1].item()), momentum)

  def momentum_flat(self, p, g, acc, mask, n_decay, wd, g_scale, lr, momentum):
    ge, pn = self._eff_grad(p, g, mask, n_decay, wd, g_scale)
    an = acc.numpy()
    an[:] = np.float32(momentum) * an + ge
    pn -= np.float32(lr) * an

  # -- weight sparsification / channel pruning ---------------------------------------------------------------------------------
  def ws_bkup_merge_abs(self, var, bkup, mask, abs_out):
    bkup.copy_(torch.where(mask > 0.5, var, bkup))
    abs_out.copy_(bkup.abs())

  def kth_largest_nonneg(self, a, k_desc_index, out, workspace):
    out[0] = torch.sort(a, descending=True).values[int(k_desc_index)]

  def ws_mask_apply(self, var, bkup, mask, thr):
    mask.copy_((bkup.abs() > thr[0]).float())
    var.copy_(bkup * mask)

  def count_nonzero(self, x, out_u64):
    out_u64 += int(torch.count_nonzero(x))

  def cp_build_mask(self, mask, keep_in, keep_out, Oc, RS, I):
    m = mask.view(Oc, RS, I)
    m.fill_(1.0)
    m[:, :, ~keep_in.bool()] = 0
    m[~keep_out.bool()] = 0

  # -- input pipeline tail ----------------------------------------------------------------------------------------------
  def image_resize_bilinear(self, src_u8, desc_u8, out, mean):
    """pf_image_resize_bilinear, element by element from the PfImageDesc semantics (include/pocketflow_hip.h)."""
    table = np.frombuffer(desc_u8.cpu().numpy().tobytes(), dtype=self.IMAGE_DESC_DTYPE)
    src = src_u8.cpu().numpy()
    B, OH, OW, _ = out.shape
    res = np.zeros((B, OH, OW, 3), np.float32)
    for b, d in enumerate(table):
      h, w = int(d['h']), int(d['w'])
      img = src[int(d['offset']):int(d['offset']) + h * w * 3].reshape(h, w, 3).astype(np.float32)
      if d['flip']:
        img = img[:, ::-1]
      for y in range(OH):
        iy = np.float32(y + int(d['off_y'])) * np.float32(d['scale_y'])
        y0 = int(np.floor(iy)); y1 = min(int(np.ceil(iy)), h - 1); ly = np.float32(iy - np.floor(iy))
        ix = (np.arange(OW) + int(d['off_x'])).astype(np.float32) * np.float32(d['scale_x'])
        x0 = np.floor(ix).astype(np.int64); x1 = np.minimum(np.ceil(ix).astype(np.int64), w - 1)
        lx = (ix - np.floor(ix)).astype(np.float32)[:, None]
        top = img[y0, x0] + (img[y0, x1] - img[y0, x0]) * lx
        bot = img[y1, x0] + (img[y1, x1] - img[y1, x0]) * lx
        res[b, y] = (top + (bot - top) * ly) - np.asarray(mean, np.float32)
    out.copy_(torch.from_numpy(res).to(out.dtype))
