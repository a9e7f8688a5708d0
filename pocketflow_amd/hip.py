"""ctypes binding of the gfx950 kernel library (include/pocketflow_hip.h).

This is the ONLY way the Python host reaches the device code: raw device pointers
(``tensor.data_ptr()``), sizes and the current HIP stream go through the C ABI -- no torch types
cross the boundary.  There is no CPU fallback: if the shared library is missing the import of
this module raises, and every wrapper raises on a non-zero hipError_t.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

import numpy as np
import torch

# PF_HIP_LIB: a complete variant build of the library (tools/gpu/build_variant.sh) for A/B runs in one GPU box; unset = the in-tree one
_LIB_PATH = os.environ.get('PF_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libpocketflow_hip.so')

PF_F32, PF_BF16 = 0, 1
PF_ACT_NONE, PF_ACT_RELU, PF_ACT_RELU6 = 0, 1, 2
PF_BUCKET_TENSOR, PF_BUCKET_CHANNEL, PF_BUCKET_SPLIT = 0, 1, 2
PF_CHUNK = 4096

# numpy mirrors of the C structs (checked against sizeof in tests)
SEG_DTYPE = np.dtype([('offset', '<i8'), ('len', '<i8'), ('RS', '<i4'), ('layout', '<i4'),
                      ('I', '<i4'), ('O', '<i4'), ('mode', '<i4'), ('bits', '<i4'),
                      ('bucket_size', '<i4'), ('n_bucket', '<i4'), ('slot_offset', '<i8'),
                      ('cb_offset', '<i8')])
BLOCK_DTYPE = np.dtype([('seg', '<i4'), ('chunk', '<i4'), ('row0', '<i4'), ('nrows', '<i4')])
assert SEG_DTYPE.itemsize == 64 and BLOCK_DTYPE.itemsize == 16

ACT_CODES = {None: PF_ACT_NONE, 'none': PF_ACT_NONE, 'Relu': PF_ACT_RELU, 'relu': PF_ACT_RELU,
             'Relu6': PF_ACT_RELU6, 'relu6': PF_ACT_RELU6}

# every symbol include/pocketflow_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    'pf_error_string', 'pf_version', 'pf_minmax_slots_init', 'pf_minmax_tensor', 'pf_minmax_decode',
    'pf_uq_apply', 'pf_act_grad', 'pf_seg_minmax', 'pf_seg_uq_apply', 'pf_seg_nuq_apply',
    'pf_seg_nuq_codebook_grad', 'pf_seg_normalize', 'pf_ws_bkup_merge_abs', 'pf_kth_largest_nonneg',
    'pf_ws_mask_apply', 'pf_count_nonzero', 'pf_cp_build_mask', 'pf_cp_mask_grad', 'pf_adam_flat',
    'pf_momentum_flat', 'pf_ce_distill_fwd_bwd', 'pf_bn_stats', 'pf_bn_finalize',
    'pf_bn_act_quant_apply', 'pf_bn_bwd_stats', 'pf_bn_bwd_finalize', 'pf_bn_bwd_apply', 'pf_bn_bwd_apply_add',
    'pf_bn_eval_scale_shift',
    'pf_conv1x1_stats_groups', 'pf_conv1x1_stats_groups_k', 'pf_conv1x1_fwd', 'pf_conv1x1_bwd_data_bnstats', 'pf_conv1x1_wrw_splits',
    'pf_conv1x1_wrw', 'pf_conv2d_stats_groups', 'pf_conv2d_fwd', 'pf_conv2d_wrw_splits', 'pf_conv2d_wrw', 'pf_maxpool_fwd', 'pf_maxpool_bwd', 'pf_seg_transpose', 'pf_conv_stem_supported', 'pf_conv_stem_fwd', 'pf_conv_stem_wrw_slabs', 'pf_conv_stem_wrw',
    'pf_image_resize_bilinear', 'pf_depthwise_supported', 'pf_depthwise_groups', 'pf_depthwise_fwd', 'pf_depthwise_bwd_data',
    'pf_depthwise_wrw', 'pf_conv2d_stats_groups_geom', 'pf_tuning_reload',
    'pf_adam_flat_dev', 'pf_momentum_flat_dev', 'pf_set_floats',
    'pf_convg_fwd', 'pf_convg_bwd_data', 'pf_convg_small_splits', 'pf_convg_wrw_splits', 'pf_convg_wrw', 'pf_conv2d_bwd_data_strided',
    'pf_prox_groups', 'pf_prox_norms', 'pf_prox_apply', 'pf_im2col', 'pf_col2im',
    'pf_conv_stem3_supported', 'pf_conv_stem3_fwd', 'pf_conv_stem3_wrw_slabs', 'pf_conv_stem3_wrw',
    'pf_conv1x1_fwd_affine', 'pf_conv2d_fwd_affine', 'pf_conv2d_bwd_data_strided_stats_groups',
    'pf_conv2d_bwd_data_strided_bnstats',
]


class HipLibraryMissing(RuntimeError):
  pass


def lib_path() -> str:
  return _LIB_PATH


def _load() -> ctypes.CDLL:
  if not os.path.exists(_LIB_PATH):
    raise HipLibraryMissing(
        'pocketflow_amd: %s not found -- build it with `python -c "import __graft_entry__ as g; '
        'g.build()"` or pocketflow_amd/csrc/build.sh.  There is no CPU fallback.' % _LIB_PATH)
  lib = ctypes.CDLL(_LIB_PATH)
  lib.pf_error_string.restype = ctypes.c_char_p
  lib.pf_error_string.argtypes = [c_int]
  lib.pf_version.restype = c_int
  for name in SYMBOLS:
    fn = getattr(lib, name)            # AttributeError here = header/library mismatch
    if name not in ('pf_error_string',):
      fn.restype = c_int
  return lib


_lib = _load()


def _check(err: int, what: str) -> None:
  if err != 0:
    raise RuntimeError('%s failed: hipError %d (%s)' % (what, err, _lib.pf_error_string(err).decode()))


# torch.cuda.current_stream() builds a Stream object behind three device-index helpers: 2.8 us per launch, 1.6 ms of the 11.6 ms the
# host needs to submit a ResNet-50 step (tools/gpu/host_overhead.py).  The raw getter returns the same hipStream_t in 0.07 us.
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_get_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream() -> c_void_p:
  if _raw_stream is not None and _get_device is not None:
    return c_void_p(_raw_stream(_get_device()))
  return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t) -> c_void_p:
  if t is None:
    return c_void_p(0)
  return c_void_p(t.data_ptr())


def dtype_code(t: torch.Tensor) -> int:
  if t.dtype == torch.float32:
    return PF_F32
  if t.dtype == torch.bfloat16:
    return PF_BF16
  raise TypeError('unsupported dtype %s (float32 / bfloat16 only)' % t.dtype)


def _dev(t: torch.Tensor) -> None:
  if not t.is_cuda:
    raise RuntimeError('pocketflow_amd kernels run on the GPU only (got a %s tensor)' % t.device)


def version() -> int:
  return int(_lib.pf_version())


def tuning_reload() -> None:
  """The launchers read their PF_* tuning switches once; tools / tests that change one in-process call this afterwards."""
  _check(_lib.pf_tuning_reload(), 'pf_tuning_reload')


# ------------------------------------------------------------------------------------------------
# K1 / K2 / K4
# ------------------------------------------------------------------------------------------------

def minmax_slots_init(slots: torch.Tensor) -> None:
  _dev(slots)
  _check(_lib.pf_minmax_slots_init(_ptr(slots), c_int64(slots.numel() // 2), _stream()), 'pf_minmax_slots_init')


def minmax_tensor(x: torch.Tensor, slot: torch.Tensor, act=None) -> None:
  _dev(x)
  _check(_lib.pf_minmax_tensor(_ptr(x), c_int64(x.numel()), c_int(dtype_code(x)), c_int(ACT_CODES[act]),
                               _ptr(slot), _stream()), 'pf_minmax_tensor')


def minmax_decode(slots: torch.Tensor) -> torch.Tensor:
  n = slots.numel() // 2
  out = torch.empty((n, 2), dtype=torch.float32, device=slots.device)
  _check(_lib.pf_minmax_decode(_ptr(slots), c_int64(n), _ptr(out), _stream()), 'pf_minmax_decode')
  return out


def uq_apply(x: torch.Tensor, y: torch.Tensor, slot: torch.Tensor, bits: int, act=None) -> None:
  _dev(x)
  _check(_lib.pf_uq_apply(_ptr(x), _ptr(y), c_int64(x.numel()), c_int(dtype_code(x)), c_int(dtype_code(y)),
                          c_int(ACT_CODES[act]), _ptr(slot), c_int(int(bits)), _stream()), 'pf_uq_apply')


def act_grad(g: torch.Tensor, u: torch.Tensor, dx: torch.Tensor, act) -> None:
  _dev(g)
  assert g.dtype == u.dtype == dx.dtype
  _check(_lib.pf_act_grad(_ptr(g), _ptr(u), _ptr(dx), c_int64(g.numel()), c_int(dtype_code(g)),
                          c_int(ACT_CODES[act]), _stream()), 'pf_act_grad')


# ------------------------------------------------------------------------------------------------
# segment (all-weights) kernels
# ------------------------------------------------------------------------------------------------

def seg_minmax(w_flat, segs, blocks, n_blocks, slots) -> None:
  _check(_lib.pf_seg_minmax(_ptr(w_flat), _ptr(segs), _ptr(blocks), c_int(n_blocks), _ptr(slots), _stream()),
         'pf_seg_minmax')


def seg_uq_apply(w_flat, qw_flat, segs, blocks, n_blocks, slots) -> None:
  _check(_lib.pf_seg_uq_apply(_ptr(w_flat), _ptr(qw_flat), c_int(dtype_code(qw_flat)), _ptr(segs), _ptr(blocks),
                              c_int(n_blocks), _ptr(slots), _stream()), 'pf_seg_uq_apply')


def seg_nuq_apply(w_flat, qw_flat, idx_flat, codebooks, segs, blocks, n_blocks, slots) -> None:
  _check(_lib.pf_seg_nuq_apply(_ptr(w_flat), _ptr(qw_flat), c_int(dtype_code(qw_flat)), _ptr(idx_flat),
                               _ptr(codebooks), _ptr(segs), _ptr(blocks), c_int(n_blocks), _ptr(slots),
                               _stream()), 'pf_seg_nuq_apply')


def seg_nuq_codebook_grad(g_flat, idx_flat, dcodebooks, acc_ws, segs, blocks, n_blocks, slots) -> None:
  """acc_ws: int64 scratch with dcodebooks.numel() elements (fixed-point accumulators; deterministic sums)."""
  assert acc_ws.dtype == torch.int64 and acc_ws.numel() >= dcodebooks.numel()
  _check(_lib.pf_seg_nuq_codebook_grad(_ptr(g_flat), c_int(dtype_code(g_flat)), _ptr(idx_flat), _ptr(dcodebooks),
                                       _ptr(acc_ws), c_int64(dcodebooks.numel()), _ptr(segs), _ptr(blocks),
                                       c_int(n_blocks), _ptr(slots), _stream()),
         'pf_seg_nuq_codebook_grad')


def seg_normalize(w_flat, xn_out, segs, seg_index, slots) -> None:
  _check(_lib.pf_seg_normalize(_ptr(w_flat), _ptr(xn_out), _ptr(segs), c_int(seg_index), _ptr(slots), _stream()),
         'pf_seg_normalize')


# ------------------------------------------------------------------------------------------------
# weight sparsification / channel pruning
# ------------------------------------------------------------------------------------------------

def ws_bkup_merge_abs(var, bkup, mask, abs_out) -> None:
  _check(_lib.pf_ws_bkup_merge_abs(_ptr(var), _ptr(bkup), _ptr(mask), _ptr(abs_out), c_int64(var.numel()),
                                   _stream()), 'pf_ws_bkup_merge_abs')


def kth_largest_nonneg(a, k_desc_index: int, out, workspace) -> None:
  _check(_lib.pf_kth_largest_nonneg(_ptr(a), c_int64(a.numel()), c_int64(int(k_desc_index)), _ptr(out),
                                    _ptr(workspace), _stream()), 'pf_kth_largest_nonneg')


def ws_mask_apply(var, bkup, mask, thr) -> None:
  _check(_lib.pf_ws_mask_apply(_ptr(var), _ptr(bkup), _ptr(mask), _ptr(thr), c_int64(var.numel()), _stream()),
         'pf_ws_mask_apply')


def count_nonzero(x, out_u64) -> None:
  _check(_lib.pf_count_nonzero(_ptr(x), c_int64(x.numel()), _ptr(out_u64), _stream()), 'pf_count_nonzero')


def cp_build_mask(mask, keep_in, keep_out, O: int, RS: int, I: int) -> None:
  _check(_lib.pf_cp_build_mask(_ptr(mask), _ptr(keep_in), _ptr(keep_out), c_int(O), c_int(RS), c_int(I),
                               _stream()), 'pf_cp_build_mask')


def cp_mask_grad(g, keep_in, keep_out, O: int, RS: int, I: int) -> None:
  _check(_lib.pf_cp_mask_grad(_ptr(g), _ptr(keep_in), _ptr(keep_out), c_int(O), c_int(RS), c_int(I), _stream()),
         'pf_cp_mask_grad')


# ------------------------------------------------------------------------------------------------
# optimisers
# ------------------------------------------------------------------------------------------------

def adam_flat(p, g, m, v, mask, n_decay: int, wd: float, g_scale: float, lr: float, beta1: float,
              beta2: float, eps: float, beta1_power: float, beta2_power: float) -> None:
  _check(_lib.pf_adam_flat(_ptr(p), _ptr(g), c_int(dtype_code(g)), _ptr(m), _ptr(v), _ptr(mask),
                           c_int64(p.numel()), c_int64(int(n_decay)), c_float(wd), c_float(g_scale), c_float(lr),
                           c_float(beta1), c_float(beta2), c_float(eps), c_float(beta1_power),
                           c_float(beta2_power), _stream()), 'pf_adam_flat')


def momentum_flat(p, g, acc, mask, n_decay: int, wd: float, g_scale: float, lr: float,
                  momentum: float) -> None:
  _check(_lib.pf_momentum_flat(_ptr(p), _ptr(g), c_int(dtype_code(g)), _ptr(acc), _ptr(mask), c_int64(p.numel()),
                               c_int64(int(n_decay)), c_float(wd), c_float(g_scale), c_float(lr),
                               c_float(momentum), _stream()), 'pf_momentum_flat')


def adam_flat_dev(p, g, m, v, mask, n_decay: int, wd: float, g_scale: float, hp, beta1: float, beta2: float,
                  eps: float) -> None:
  """pf_adam_flat with alpha_t read from hp[0] (device float32[4]): the launch a captured step records."""
  _check(_lib.pf_adam_flat_dev(_ptr(p), _ptr(g), c_int(dtype_code(g)), _ptr(m), _ptr(v), _ptr(mask),
                               c_int64(p.numel()), c_int64(int(n_decay)), c_float(wd), c_float(g_scale), _ptr(hp),
                               c_float(beta1), c_float(beta2), c_float(eps), _stream()), 'pf_adam_flat_dev')


def momentum_flat_dev(p, g, acc, mask, n_decay: int, wd: float, g_scale: float, hp, momentum: float) -> None:
  _check(_lib.pf_momentum_flat_dev(_ptr(p), _ptr(g), c_int(dtype_code(g)), _ptr(acc), _ptr(mask), c_int64(p.numel()),
                                   c_int64(int(n_decay)), c_float(wd), c_float(g_scale), _ptr(hp),
                                   c_float(momentum), _stream()), 'pf_momentum_flat_dev')


def set_floats(dst, a: float, b: float = 0.0, c: float = 0.0, d: float = 0.0) -> None:
  """dst[0..3] = (a, b, c, d) by a one-thread kernel whose arguments carry the values."""
  assert dst.dtype == torch.float32 and dst.numel() >= 4
  _check(_lib.pf_set_floats(_ptr(dst), c_float(a), c_float(b), c_float(c), c_float(d), _stream()), 'pf_set_floats')


def prox_groups(rows: int, I: int) -> int:
  return int(_lib.pf_prox_groups(c_int64(rows), c_int(I)))


def prox_norms(w, g, lr: float, rows: int, I: int, partial, norms) -> None:
  """norms[I] = per-input-channel L2 norm of W - lr * G, W a float32 [rows][I] matrix (KRSC kernel), G float32 / bf16."""
  _dev(w)
  if w.dtype != torch.float32 or partial.numel() < prox_groups(rows, I) * I or norms.numel() < I:
    raise TypeError('prox_norms: float32 master kernel, workspace of prox_groups(rows, I) * I floats')
  _check(_lib.pf_prox_norms(_ptr(w), _ptr(g), c_int(dtype_code(g)), c_float(lr), c_int64(rows), c_int(I), _ptr(partial), _ptr(norms),
                            _stream()), 'pf_prox_norms')


def prox_apply(w, g, lr: float, rows: int, I: int, norms, thr) -> None:
  """W <- (W - lr * G) * max(1 - thr[0] / norms[c], 0), in place."""
  _dev(w)
  _check(_lib.pf_prox_apply(_ptr(w), _ptr(g), c_int(dtype_code(g)), c_float(lr), c_int64(rows), c_int(I), _ptr(norms), _ptr(thr),
                            _stream()), 'pf_prox_apply')


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------

def ce_distill_fwd_bwd(z_s, labels, z_t, tempr: float, loss_w: float, losses, dz_s, row_ws) -> None:
  B, C = z_s.shape
  _check(_lib.pf_ce_distill_fwd_bwd(_ptr(z_s), c_int(dtype_code(z_s)), _ptr(labels), _ptr(z_t),
                                    c_int(dtype_code(z_t) if z_t is not None else 0), c_int(B), c_int(C),
                                    c_float(tempr), c_float(loss_w), _ptr(losses), _ptr(dz_s),
                                    c_int(dtype_code(dz_s)), _ptr(row_ws), _stream()), 'pf_ce_distill_fwd_bwd')


# ------------------------------------------------------------------------------------------------
# fused BN + act + fake-quant
# ------------------------------------------------------------------------------------------------

def bn_stats(x, rows: int, C: int, partial, n_blocks: int) -> None:
  _check(_lib.pf_bn_stats(_ptr(x), c_int(dtype_code(x)), c_int64(rows), c_int(C), _ptr(partial), c_int(n_blocks),
                          _stream()), 'pf_bn_stats')


def bn_finalize(partial, n_blocks, rows, C, x, gamma, beta, moving_mean, moving_var, momentum, eps,
                training: bool, act, scale_shift, mean_invstd, slot) -> None:
  _check(_lib.pf_bn_finalize(_ptr(partial), c_int(n_blocks), c_int64(rows), c_int(C), _ptr(x),
                             c_int(dtype_code(x)), _ptr(gamma), _ptr(beta), _ptr(moving_mean), _ptr(moving_var),
                             c_float(momentum), c_float(eps), c_int(1 if training else 0), c_int(ACT_CODES[act]),
                             _ptr(scale_shift), _ptr(mean_invstd), _ptr(slot), _stream()), 'pf_bn_finalize')


def bn_act_quant_apply(x, q, rows, C, scale_shift, act, slot, bits: int, quantize: bool) -> None:
  _check(_lib.pf_bn_act_quant_apply(_ptr(x), _ptr(q), c_int(dtype_code(x)), c_int64(rows), c_int(C),
                                    _ptr(scale_shift), c_int(ACT_CODES[act]), _ptr(slot), c_int(int(bits)),
                                    c_int(1 if quantize else 0), _stream()), 'pf_bn_act_quant_apply')


def bn_bwd_stats(dq, x, rows, C, scale_shift, mean_invstd, act, partial, n_blocks) -> None:
  _check(_lib.pf_bn_bwd_stats(_ptr(dq), _ptr(x), c_int(dtype_code(x)), c_int64(rows), c_int(C), _ptr(scale_shift),
                              _ptr(mean_invstd), c_int(ACT_CODES[act]), _ptr(partial), c_int(n_blocks),
                              _stream()), 'pf_bn_bwd_stats')


def bn_bwd_finalize(partial, n_blocks, C, dgamma, dbeta) -> None:
  _check(_lib.pf_bn_bwd_finalize(_ptr(partial), c_int(n_blocks), c_int(C), _ptr(dgamma), _ptr(dbeta), _stream()),
         'pf_bn_bwd_finalize')


def bn_bwd_apply(dq, x, dx, rows, C, scale_shift, mean_invstd, dgamma, dbeta, act, addend=None) -> None:
  _check(_lib.pf_bn_bwd_apply_add(_ptr(dq), _ptr(x), _ptr(addend), _ptr(dx), c_int(dtype_code(x)), c_int64(rows),
                                  c_int(C), _ptr(scale_shift), _ptr(mean_invstd), _ptr(dgamma), _ptr(dbeta),
                                  c_int(ACT_CODES[act]), _stream()), 'pf_bn_bwd_apply_add')


def bn_eval_scale_shift(gamma, beta, moving_mean, moving_var, eps: float, scale_shift) -> None:
  _check(_lib.pf_bn_eval_scale_shift(_ptr(gamma), _ptr(beta), _ptr(moving_mean), _ptr(moving_var), c_float(eps),
                                     c_int(gamma.numel()), _ptr(scale_shift), _stream()), 'pf_bn_eval_scale_shift')


# ------------------------------------------------------------------------------------------------
# fused 1x1 convolutions
# ------------------------------------------------------------------------------------------------

def conv1x1_stats_groups(M: int, N: int, K: int, prologue: bool = False) -> int:
  """Rows of the partial-statistics array of an [M][K] x [N][K] convolution (depends on the kernel variant the shape is
  dispatched to; `prologue`: the call passes scale_shift)."""
  return int(_lib.pf_conv1x1_stats_groups_k(c_int(M), c_int(N), c_int(K), c_int(1 if prologue else 0)))


def conv1x1_wrw_splits(M: int, N: int, K: int) -> int:
  return int(_lib.pf_conv1x1_wrw_splits(c_int(M), c_int(N), c_int(K)))


def conv1x1_fwd(X, W, Y, M: int, N: int, K: int, R=None, scale_shift=None, act=None, slot=None, bits: int = 8,
                partial=None, geom=None, ymap: bool = False, out_scale_shift=None, out_act=None) -> None:
  """geom = (Ho, Wo, H, Wd, stride) for a strided 1x1 convolution, None for stride 1.  out_scale_shift ([2][N] float32) / out_act:
  the consumer's inference-mode BN + activation folded into the epilogue (pf_conv1x1_fwd_affine; no residual / statistics / slot)."""
  _dev(X)
  Ho, Wo, H, Wd, stride = geom if geom is not None else (0, 0, 0, 0, 1)
  if out_scale_shift is not None:
    if R is not None or partial is not None or slot is not None or ymap:
      raise ValueError('the folded output pass takes no residual, statistics, quantiser or row map')
    _check(_lib.pf_conv1x1_fwd_affine(_ptr(X), _ptr(W), _ptr(Y), _ptr(scale_shift), c_int(ACT_CODES[act]), _ptr(out_scale_shift),
                                      c_int(ACT_CODES[out_act]), c_int(M), c_int(N), c_int(K), c_int(Ho), c_int(Wo), c_int(H),
                                      c_int(Wd), c_int(stride), _stream()), 'pf_conv1x1_fwd_affine')
    return
  _check(_lib.pf_conv1x1_fwd(_ptr(X), _ptr(W), _ptr(Y), _ptr(R), _ptr(scale_shift), c_int(ACT_CODES[act]),
                             _ptr(slot), c_int(int(bits)), _ptr(partial), c_int(M), c_int(N), c_int(K), c_int(Ho),
                             c_int(Wo), c_int(H), c_int(Wd), c_int(stride), c_int(1 if ymap else 0), _stream()),
         'pf_conv1x1_fwd')


def conv1x1_wrw(dY, X, dW, workspace, M: int, N: int, K: int, scale_shift=None, act=None, slot=None,
                bits: int = 8, geom=None) -> None:
  _dev(X)
  Ho, Wo, H, Wd, stride = geom if geom is not None else (0, 0, 0, 0, 1)
  _check(_lib.pf_conv1x1_wrw(_ptr(dY), _ptr(X), _ptr(dW), c_int(dtype_code(dW)), _ptr(workspace), _ptr(scale_shift),
                             c_int(ACT_CODES[act]), _ptr(slot), c_int(int(bits)), c_int(M), c_int(N), c_int(K),
                             c_int(Ho), c_int(Wo), c_int(H), c_int(Wd), c_int(stride), _stream()), 'pf_conv1x1_wrw')


def conv1x1_bwd_data_bnstats(dY, Wt, dQ, bn_x, bn_scale_shift, bn_mean_invstd, bn_act, partial, M: int, N: int,
                             K: int) -> None:
  _dev(dY)
  _check(_lib.pf_conv1x1_bwd_data_bnstats(_ptr(dY), _ptr(Wt), _ptr(dQ), _ptr(bn_x), _ptr(bn_scale_shift),
                                          _ptr(bn_mean_invstd), c_int(ACT_CODES[bn_act]), _ptr(partial), c_int(M),
                                          c_int(N), c_int(K), _stream()), 'pf_conv1x1_bwd_data_bnstats')


# ------------------------------------------------------------------------------------------------
# RxS convolutions as implicit GEMMs
# ------------------------------------------------------------------------------------------------

_zero_page = {}


def zero_page(device):
  """128+ zero bytes on `device`: the source of padding taps of pf_conv2d_fwd."""
  key = str(device)
  z = _zero_page.get(key)
  if z is None:
    z = _zero_page[key] = torch.zeros(256, dtype=torch.bfloat16, device=device)
  return z


def conv2d_stats_groups(M: int, N: int, geom=None) -> int:
  """Rows of the statistics array of a pf_conv2d_fwd call.  `geom` = (imgs, H, Wd, C, N, th, tw, stride, pad_h, pad_w, Ho, Wo)
  of THAT call: the kernel (and with it the number of workgroup rows) depends on the geometry."""
  if geom is None:
    # the (M, N) query answers for a 1x1 product only; which kernel an R x S launch gets -- and with it the number of
    # statistics rows -- depends on its window, channel counts and image size (pf_conv3x3_c64.hip since round 6)
    raise ValueError('conv2d_stats_groups: pass geom = (imgs, H, Wd, C, N, th, tw, stride, pad_h, pad_w, Ho, Wo) of the pf_conv2d_fwd call')
  return int(_lib.pf_conv2d_stats_groups_geom(*[c_int(int(v)) for v in geom]))


def conv2d_fwd(X, W, Y, imgs: int, H: int, Wd: int, C: int, N: int, th: int, tw: int, stride: int, pad_h: int,
               pad_w: int, Ho: int, Wo: int, R=None, partial=None, bn_x=None, bn_scale_shift=None,
               bn_mean_invstd=None, bn_act=None, out_scale_shift=None, out_act=None) -> None:
  """X: NHWC memory [imgs][H][Wd][C] bf16, W: KRSC memory [N][th][tw][C] bf16, Y: [imgs][Ho][Wo][N] bf16.
  out_scale_shift / out_act: the consumer's inference-mode BN + activation folded into the epilogue (pf_conv2d_fwd_affine)."""
  _dev(X)
  if out_scale_shift is not None:
    if R is not None or partial is not None or bn_x is not None:
      raise ValueError('the folded output pass takes no residual or statistics')
    _check(_lib.pf_conv2d_fwd_affine(_ptr(X), _ptr(W), _ptr(Y), _ptr(zero_page(X.device)), _ptr(out_scale_shift),
                                     c_int(ACT_CODES[out_act]), c_int(imgs), c_int(H), c_int(Wd), c_int(C), c_int(N), c_int(th),
                                     c_int(tw), c_int(stride), c_int(pad_h), c_int(pad_w), c_int(Ho), c_int(Wo), _stream()),
           'pf_conv2d_fwd_affine')
    return
  _check(_lib.pf_conv2d_fwd(_ptr(X), _ptr(W), _ptr(Y), _ptr(zero_page(X.device)), _ptr(R), _ptr(partial), _ptr(bn_x),
                            _ptr(bn_scale_shift), _ptr(bn_mean_invstd), c_int(ACT_CODES[bn_act]), c_int(imgs), c_int(H),
                            c_int(Wd), c_int(C), c_int(N), c_int(th), c_int(tw), c_int(stride), c_int(pad_h),
                            c_int(pad_w), c_int(Ho), c_int(Wo), _stream()), 'pf_conv2d_fwd')


def conv2d_wrw_splits(M: int, N: int, C: int, taps: int) -> int:
  return int(_lib.pf_conv2d_wrw_splits(c_int(M), c_int(N), c_int(C), c_int(taps)))


def conv2d_wrw(dY, X, dW, workspace, imgs: int, H: int, Wd: int, C: int, N: int, th: int, tw: int, stride: int,
               pad_h: int, pad_w: int, Ho: int, Wo: int) -> None:
  """dW: KRSC memory [N][th][tw][C], float32 or bf16."""
  _dev(X)
  _check(_lib.pf_conv2d_wrw(_ptr(dY), _ptr(X), _ptr(dW), c_int(dtype_code(dW)), _ptr(workspace), c_int(imgs), c_int(H),
                            c_int(Wd), c_int(C), c_int(N), c_int(th), c_int(tw), c_int(stride), c_int(pad_h),
                            c_int(pad_w), c_int(Ho), c_int(Wo), _stream()), 'pf_conv2d_wrw')


def depthwise_supported(C: int, k: int, stride: int) -> bool:
  return bool(_lib.pf_depthwise_supported(c_int(C), c_int(k), c_int(stride)))


def depthwise_groups(B: int, Ho: int, Wo: int, C: int) -> int:
  return int(_lib.pf_depthwise_groups(c_int(B), c_int(Ho), c_int(Wo), c_int(C)))


def depthwise_fwd(X, W, Y, B: int, H: int, Wd: int, C: int, k: int, stride: int, pad_h: int, pad_w: int, Ho: int, Wo: int,
                  partial=None) -> None:
  """X: NHWC memory [B][H][Wd][C], W: [C][k][k] in X's dtype, Y: [B][Ho][Wo][C]; partial: [G][4][C] float32 or None."""
  _dev(X)
  _check(_lib.pf_depthwise_fwd(_ptr(X), _ptr(W), _ptr(Y), c_int(dtype_code(X)), _ptr(partial), c_int(B), c_int(H), c_int(Wd),
                               c_int(C), c_int(k), c_int(stride), c_int(pad_h), c_int(pad_w), c_int(Ho), c_int(Wo),
                               _stream()), 'pf_depthwise_fwd')


def depthwise_bwd_data(dY, W, dX, B: int, H: int, Wd: int, C: int, k: int, stride: int, pad_h: int, pad_w: int, Ho: int,
                       Wo: int) -> None:
  _dev(dY)
  _check(_lib.pf_depthwise_bwd_data(_ptr(dY), _ptr(W), _ptr(dX), c_int(dtype_code(dY)), c_int(B), c_int(H), c_int(Wd), c_int(C),
                                    c_int(k), c_int(stride), c_int(pad_h), c_int(pad_w), c_int(Ho), c_int(Wo), _stream()),
         'pf_depthwise_bwd_data')


def depthwise_wrw(dY, X, dW, slabs, B: int, H: int, Wd: int, C: int, k: int, stride: int, pad_h: int, pad_w: int, Ho: int,
                  Wo: int) -> None:
  """dW: [C][k][k], float32 or bf16; slabs: (depthwise_groups(B, Ho, Wo, C) + 32) * C * k * k floats."""
  _dev(X)
  if slabs.dtype != torch.float32 or slabs.numel() < (depthwise_groups(B, Ho, Wo, C) + 32) * C * k * k:
    raise TypeError('depthwise_wrw: float32 workspace of (groups + 32) * C * k * k elements')
  _check(_lib.pf_depthwise_wrw(_ptr(dY), _ptr(X), _ptr(dW), c_int(dtype_code(X)), c_int(dtype_code(dW)), _ptr(slabs), c_int(B),
                               c_int(H), c_int(Wd), c_int(C), c_int(k), c_int(stride), c_int(pad_h), c_int(pad_w), c_int(Ho),
                               c_int(Wo), _stream()), 'pf_depthwise_wrw')


def conv2d_bwd_data_strided_stats_groups(B: int, H: int, Wd: int, C: int, stride: int) -> int:
  return int(_lib.pf_conv2d_bwd_data_strided_stats_groups(c_int(B), c_int(H), c_int(Wd), c_int(C), c_int(stride)))


def conv2d_bwd_data_strided(dY, Wt, dX, B: int, H: int, Wd: int, C: int, N: int, R: int, S: int, stride: int, pad_h: int,
                            pad_w: int, Ho: int, Wo: int, partial=None, bn_x=None, bn_scale_shift=None, bn_mean_invstd=None,
                            bn_act=None) -> None:
  """dX[B][H][Wd][C] of a strided convolution from dY[B][Ho][Wo][N] and the flipped / transposed kernel Wt[C][R][S][N] (bf16):
  stride * stride launches of the implicit-GEMM kernel, one per output-parity class (pf_igemm.hip).  bn_x ...: the BN-backward
  sums of the BN in front of the convolution in the epilogues (partial [conv2d_bwd_data_strided_stats_groups(...)][2][C])."""
  _dev(dY)
  if dY.dtype != torch.bfloat16 or Wt.dtype != torch.bfloat16 or dX.dtype != torch.bfloat16:
    raise TypeError('conv2d_bwd_data_strided: bf16 tensors')
  if bn_x is not None:
    _check(_lib.pf_conv2d_bwd_data_strided_bnstats(_ptr(dY), _ptr(Wt), _ptr(dX), _ptr(zero_page(dY.device)), _ptr(bn_x),
                                                   _ptr(bn_scale_shift), _ptr(bn_mean_invstd), c_int(ACT_CODES[bn_act]),
                                                   _ptr(partial), c_int(B), c_int(H), c_int(Wd), c_int(C), c_int(N), c_int(R),
                                                   c_int(S), c_int(stride), c_int(pad_h), c_int(pad_w), c_int(Ho), c_int(Wo),
                                                   _stream()), 'pf_conv2d_bwd_data_strided_bnstats')
    return
  _check(_lib.pf_conv2d_bwd_data_strided(_ptr(dY), _ptr(Wt), _ptr(dX), _ptr(zero_page(dY.device)), c_int(B), c_int(H), c_int(Wd),
                                         c_int(C), c_int(N), c_int(R), c_int(S), c_int(stride), c_int(pad_h), c_int(pad_w), c_int(Ho),
                                         c_int(Wo), _stream()), 'pf_conv2d_bwd_data_strided')


def im2col(X, Xcol, B: int, H: int, Wd: int, C: int, R: int, S: int, stride: int, pad_h: int, pad_w: int, Ho: int, Wo: int) -> None:
  """Xcol[B*Ho*Wo][R*S*C] <- X[B][H][Wd][C] (bf16, C % 8 == 0; taps outside the image are zeros)."""
  _dev(X)
  if X.dtype != torch.bfloat16 or Xcol.dtype != torch.bfloat16 or Xcol.numel() < B * Ho * Wo * R * S * C:
    raise TypeError('im2col: bf16 tensors, Xcol of B*Ho*Wo*R*S*C elements')
  _check(_lib.pf_im2col(_ptr(X), _ptr(Xcol), c_int(B), c_int(H), c_int(Wd), c_int(C), c_int(R), c_int(S), c_int(stride), c_int(pad_h),
                        c_int(pad_w), c_int(Ho), c_int(Wo), _stream()), 'pf_im2col')


def col2im(dXcol, dX, B: int, H: int, Wd: int, C: int, R: int, S: int, stride: int, pad_h: int, pad_w: int, Ho: int, Wo: int) -> None:
  """dX[B][H][Wd][C] <- the gather-sum of dXcol[B*Ho*Wo][R*S*C] (inverse of im2col; bf16)."""
  _dev(dXcol)
  if dXcol.dtype != torch.bfloat16 or dX.dtype != torch.bfloat16:
    raise TypeError('col2im: bf16 tensors')
  _check(_lib.pf_col2im(_ptr(dXcol), _ptr(dX), c_int(B), c_int(H), c_int(Wd), c_int(C), c_int(R), c_int(S), c_int(stride), c_int(pad_h),
                        c_int(pad_w), c_int(Ho), c_int(Wo), _stream()), 'pf_col2im')


# ------------------------------------------------------------------------------------------------
# general convolutions / dense (pf_convg.hip): any shape and stride, float32 or bf16, float32 accumulation
# ------------------------------------------------------------------------------------------------

def convg_fwd(X, Wk, bias, Y, B: int, H: int, Wd: int, C: int, N: int, R: int, S: int, stride: int, pad_h: int, pad_w: int,
              Ho: int, Wo: int, slab=None) -> None:
  """Y[B][Ho][Wo][N] = conv(X[B][H][Wd][C], Wk[N][R][S][C]) (+ bias[N], float32); pad_*: begin pads."""
  _dev(X)
  if Wk.dtype != X.dtype or Y.dtype != X.dtype or (bias is not None and bias.dtype != torch.float32):
    raise TypeError('convg_fwd: X / Wk / Y share one dtype, the bias is float32')
  _check(_lib.pf_convg_fwd(_ptr(X), _ptr(Wk), _ptr(bias), _ptr(Y), c_int(dtype_code(X)), c_int(B), c_int(H), c_int(Wd), c_int(C),
                           c_int(N), c_int(R), c_int(S), c_int(stride), c_int(pad_h), c_int(pad_w), c_int(Ho), c_int(Wo), _ptr(slab),
                           c_int64(slab.numel() if slab is not None else 0), _stream()), 'pf_convg_fwd')


def convg_bwd_data(dY, Wk, dX, B: int, H: int, Wd: int, C: int, N: int, R: int, S: int, stride: int, pad_h: int, pad_w: int,
                   Ho: int, Wo: int, slab=None) -> None:
  _dev(dY)
  if Wk.dtype != dY.dtype or dX.dtype != dY.dtype:
    raise TypeError('convg_bwd_data: dY / Wk / dX share one dtype')
  _check(_lib.pf_convg_bwd_data(_ptr(dY), _ptr(Wk), _ptr(dX), c_int(dtype_code(dY)), c_int(B), c_int(H), c_int(Wd), c_int(C),
                                c_int(N), c_int(R), c_int(S), c_int(stride), c_int(pad_h), c_int(pad_w), c_int(Ho), c_int(Wo),
                                _ptr(slab), c_int64(slab.numel() if slab is not None else 0), _stream()), 'pf_convg_bwd_data')


def convg_small_splits(M: int, Nc: int, K: int) -> int:
  """Contraction splits of a forward / backward-data launch with M x Nc outputs and K terms each (0: none) -- shape only; the
  workspace to pass as `slab` is splits * M * Nc float32 elements."""
  return int(_lib.pf_convg_small_splits(c_int(M), c_int(Nc), c_int(K)))


def convg_wrw_splits(B: int, C: int, N: int, R: int, S: int, Ho: int, Wo: int) -> int:
  return int(_lib.pf_convg_wrw_splits(c_int(B), c_int(C), c_int(N), c_int(R), c_int(S), c_int(Ho), c_int(Wo)))


def convg_wrw(dY, X, dW, slab, B: int, H: int, Wd: int, C: int, N: int, R: int, S: int, stride: int, pad_h: int, pad_w: int,
              Ho: int, Wo: int) -> None:
  """dW[N][R][S][C] (float32 or bf16); slab: float32 workspace of convg_wrw_splits(...) * N * R * S * C elements."""
  _dev(X)
  if dY.dtype != X.dtype or slab.dtype != torch.float32 or slab.numel() < convg_wrw_splits(B, C, N, R, S, Ho, Wo) * N * R * S * C:
    raise TypeError('convg_wrw: dY / X share one dtype; float32 workspace of splits * N * R * S * C elements')
  _check(_lib.pf_convg_wrw(_ptr(dY), _ptr(X), _ptr(dW), c_int(dtype_code(X)), c_int(dtype_code(dW)), _ptr(slab), c_int(B), c_int(H),
                           c_int(Wd), c_int(C), c_int(N), c_int(R), c_int(S), c_int(stride), c_int(pad_h), c_int(pad_w), c_int(Ho),
                           c_int(Wo), _stream()), 'pf_convg_wrw')


TILE_DTYPE = np.dtype([('src_off', '<i8'), ('dst_off', '<i8'), ('O', '<i4'), ('I', '<i4'), ('src_ld', '<i4'),
                       ('dst_ld', '<i4'), ('o0', '<i4'), ('i0', '<i4'), ('r0', '<i4'), ('r1', '<i4')])
assert TILE_DTYPE.itemsize == 48          # struct PfTransposeTile: two int64 + eight int32


def seg_transpose(src_flat, dst_flat, tiles, n_tiles: int) -> None:
  _dev(src_flat)
  _check(_lib.pf_seg_transpose(_ptr(src_flat), _ptr(dst_flat), c_int(dtype_code(src_flat)), _ptr(tiles), c_int(n_tiles),
                               _stream()), 'pf_seg_transpose')


# ------------------------------------------------------------------------------------------------
# the ResNet stem (7x7 / 2, 3 -> 64 channels)
# ------------------------------------------------------------------------------------------------

def conv_stem_supported(H: int, Wd: int, C: int, N: int, k: int, stride: int, pad: int) -> bool:
  return bool(_lib.pf_conv_stem_supported(c_int(H), c_int(Wd), c_int(C), c_int(N), c_int(k), c_int(stride), c_int(pad)))


def conv_stem_fwd(X, W, Y, imgs: int, H: int, Wd: int) -> None:
  """X: NHWC memory [imgs][H][Wd][3] bf16, W: KRSC memory [64][7][7][3] bf16, Y: [imgs][H/2][Wd/2][64] bf16."""
  _dev(X)
  _check(_lib.pf_conv_stem_fwd(_ptr(X), _ptr(W), _ptr(Y), c_int(imgs), c_int(H), c_int(Wd), _stream()), 'pf_conv_stem_fwd')


def conv_stem_wrw_slabs(imgs: int, H: int, Wd: int) -> int:
  return int(_lib.pf_conv_stem_wrw_slabs(c_int(imgs), c_int(H), c_int(Wd)))


def conv_stem_wrw(dY, X, dW, workspace, imgs: int, H: int, Wd: int) -> None:
  """dY: [imgs][H/2][Wd/2][64] bf16, X: [imgs][H][Wd][3] bf16, dW: KRSC memory [64][7][7][3] float32 / bf16;
  workspace: float32, (conv_stem_wrw_slabs(...) + 32) * 64 * 147 elements."""
  _dev(dY)
  _check(_lib.pf_conv_stem_wrw(_ptr(dY), _ptr(X), _ptr(dW), c_int(dtype_code(dW)), _ptr(workspace), c_int(imgs), c_int(H),
                               c_int(Wd), _stream()), 'pf_conv_stem_wrw')


def conv_stem3_supported(H: int, Wd: int, C: int, N: int, k: int, stride: int, pad_h: int, pad_w: int, Ho: int, Wo: int) -> bool:
  return bool(_lib.pf_conv_stem3_supported(c_int(H), c_int(Wd), c_int(C), c_int(N), c_int(k), c_int(stride), c_int(pad_h), c_int(pad_w),
                                           c_int(Ho), c_int(Wo)))


def conv_stem3_fwd(X, W, Y, imgs: int, H: int, Wd: int, N: int, pad_h: int, pad_w: int, Ho: int, Wo: int) -> None:
  """The MobileNet stem.  X: NHWC memory [imgs][H][Wd][3] bf16, W: KRSC memory [N][3][3][3] bf16, Y: [imgs][Ho][Wo][N] bf16;
  pad_h / pad_w: FRONT pads of 'SAME'."""
  _dev(X)
  _check(_lib.pf_conv_stem3_fwd(_ptr(X), _ptr(W), _ptr(Y), c_int(imgs), c_int(H), c_int(Wd), c_int(N), c_int(pad_h), c_int(pad_w),
                                c_int(Ho), c_int(Wo), _stream()), 'pf_conv_stem3_fwd')


def conv_stem3_wrw_slabs(imgs: int, H: int, Wd: int, N: int, pad_h: int, pad_w: int, Ho: int, Wo: int) -> int:
  return int(_lib.pf_conv_stem3_wrw_slabs(c_int(imgs), c_int(H), c_int(Wd), c_int(N), c_int(pad_h), c_int(pad_w), c_int(Ho), c_int(Wo)))


def conv_stem3_wrw(dY, X, dW, workspace, imgs: int, H: int, Wd: int, N: int, pad_h: int, pad_w: int, Ho: int, Wo: int) -> None:
  """dW: KRSC memory [N][3][3][3] float32 / bf16; workspace: float32, (conv_stem3_wrw_slabs(...) + 32) * N * 27 elements."""
  _dev(dY)
  _check(_lib.pf_conv_stem3_wrw(_ptr(dY), _ptr(X), _ptr(dW), c_int(dtype_code(dW)), _ptr(workspace), c_int(imgs), c_int(H), c_int(Wd),
                                c_int(N), c_int(pad_h), c_int(pad_w), c_int(Ho), c_int(Wo), _stream()), 'pf_conv_stem3_wrw')


# ------------------------------------------------------------------------------------------------
# max-pooling
# ------------------------------------------------------------------------------------------------

def maxpool_fwd(x, y, idx, B: int, H: int, W: int, C: int, k: int, stride: int, pad_h: int, pad_w: int, Ho: int,
                Wo: int) -> None:
  _dev(x)
  _check(_lib.pf_maxpool_fwd(_ptr(x), _ptr(y), _ptr(idx), c_int(dtype_code(x)), c_int(B), c_int(H), c_int(W), c_int(C),
                             c_int(k), c_int(stride), c_int(pad_h), c_int(pad_w), c_int(Ho), c_int(Wo), _stream()),
         'pf_maxpool_fwd')


def maxpool_bwd(dy, idx, dx, B: int, H: int, W: int, C: int, k: int, stride: int, pad_h: int, pad_w: int, Ho: int,
                Wo: int) -> None:
  _dev(dy)
  _check(_lib.pf_maxpool_bwd(_ptr(dy), _ptr(idx), _ptr(dx), c_int(dtype_code(dy)), c_int(B), c_int(H), c_int(W), c_int(C),
                             c_int(k), c_int(stride), c_int(pad_h), c_int(pad_w), c_int(Ho), c_int(Wo), _stream()),
         'pf_maxpool_bwd')


# ------------------------------------------------------------------------------------------------
# input pipeline tail
# ------------------------------------------------------------------------------------------------

IMAGE_DESC_DTYPE = np.dtype([('offset', '<i8'), ('h', '<i4'), ('w', '<i4'), ('scale_y', '<f4'), ('scale_x', '<f4'),
                             ('off_y', '<i4'), ('off_x', '<i4'), ('flip', '<i4'), ('reserved', '<i4')])   # = PfImageDesc


def image_resize_bilinear(src_u8, desc_u8, out, mean) -> None:
  """out [B, OH, OW, 3] (float32 / bfloat16) <- TF-1.x bilinear resize (+ flip, window, - mean) of the packed uint8
  HWC images in `src_u8`, described by `desc_u8` (the bytes of an IMAGE_DESC_DTYPE array, on the device)."""
  B, OH, OW, C = out.shape
  assert C == 3 and out.is_contiguous() and desc_u8.numel() == B * IMAGE_DESC_DTYPE.itemsize
  _check(_lib.pf_image_resize_bilinear(_ptr(src_u8), _ptr(desc_u8), _ptr(out), c_int(dtype_code(out)), c_int(B),
                                       c_int(OH), c_int(OW), c_float(mean[0]), c_float(mean[1]), c_float(mean[2]),
                                       _stream()), 'pf_image_resize_bilinear')
