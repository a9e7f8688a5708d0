"""CPU oracle for the PocketFlow compression hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-NumPy restatement of the arithmetic of the reference's hot path
(Tencent/PocketFlow, TensorFlow-1.x graph code).  It is the *checker* for the HIP kernels in
``pocketflow_amd/csrc``; nothing under ``pocketflow_amd/`` may import it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it.

PARITY: PINNED BY REFERENCE-EXECUTED FIXTURES (TF primitives restated).  TensorFlow 1.x / Horovod cannot be
installed here and the reference tree holds no golden vectors for this path (SURVEY.md sections 4, 8c), so
the reference's OWN Python functions are executed instead: ``tests/golden/make_reference_golden.py`` lifts them
out of /root/reference with ``ast`` and runs them over ``oracle/tf_stub.py`` (a NumPy-eager stand-in for the
~40 stock TF ops they call); inputs + outputs are committed as ``tests/golden/reference_arrays.npz`` /
``reference_host.json`` and ``tests/test_oracle_golden.py`` holds this oracle to them bit for bit.  That pins
the op order, bucket reshapes, constants and network definitions against the reference's code; what stays a
restatement is the behaviour of each individual TF primitive (listed in tf_stub.py) and the gradient rules,
which are pinned by
 (1) hand-computable known-answer vectors (``tests/test_oracle_kat.py``),
 (2) cross-checks against independent torch-CPU implementations where the published semantics
     coincide (Adam, Momentum-SGD, soft-label cross-entropy, round-half-even), and
 (3) the real scikit-learn LassoLars / LinearRegression for the channel-pruning selector.
Every function cites the reference file:line it restates (paths relative to /root/reference).

Layout convention: as in the reference -- conv kernels are HWIO ``[kh, kw, cin, cout]``, dense
kernels ``[in, out]``, activations NHWC, everything float32.  All arithmetic is carried out in
float32 with one rounding per TF op (no FMA contraction), because every TF op in the reference
chain is a separate kernel producing a float32 tensor.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32


def f32(x):
  """Cast to a float32 ndarray / scalar."""
  return np.asarray(x, dtype=np.float32)


# ---------------------------------------------------------------------------------------------
# A.1  Uniform fake quantisation          learners/uniform_quantization/utils.py:163-306
# ---------------------------------------------------------------------------------------------

def uq_k(bits: int) -> np.float32:
  """k = float32(2**bits - 1)   (utils.py:184: tf.cast(2 ** mbits - 1, tf.float32), mbits int64)."""
  return np.float32(np.int64(2) ** np.int64(bits) - np.int64(1))


def scale(w: np.ndarray, axis: Optional[int]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
  """__scale (uq utils.py:201-231 == nuq utils.py:388-418).

  w_max/w_min over `axis` (None = whole tensor), alpha = max - min + 1e-10, beta = min,
  returns ((w - beta) / alpha, alpha, beta).  max/min carry no gradient (stop_gradient).
  """
  w = f32(w)
  w_max = w.max(axis=axis)
  w_min = w.min(axis=axis)
  eps = np.float32(1e-10)
  alpha = f32(f32(w_max - w_min) + eps)
  beta = f32(w_min)
  return f32(f32(w - beta) / alpha), alpha, beta


def inv_scale(w: np.ndarray, alpha: np.ndarray, beta: np.ndarray) -> np.ndarray:
  """__inv_scale (uq utils.py:233-245): alpha * w + beta  (mul then add, two roundings)."""
  return f32(f32(alpha * w) + beta)


def split_bucket(w: np.ndarray, bucket_size: int) -> Tuple[np.ndarray, int, int]:
  """__split_bucket (uq utils.py:247-275; nuq utils.py:435-462).

  Flatten (row-major in the reference's HWIO layout), pad with copies of the LAST element up to
  a multiple of bucket_size, reshape to [bucket_size, multiple].  Bucket j is therefore the
  strided set {flat[i*multiple + j]}_i, not a contiguous run.
  """
  flat = f32(w).reshape(-1)
  num_w = flat.shape[0]
  multiple, rest = divmod(num_w, bucket_size)
  if rest != 0:
    flat = np.concatenate([flat, np.ones(bucket_size - rest, dtype=np.float32) * flat[-1]])
    multiple += 1
  padded_num = (bucket_size - rest) if rest != 0 else 0
  return flat.reshape(bucket_size, -1), multiple, padded_num


def channel_bucket(w: np.ndarray) -> Tuple[np.ndarray, int, int]:
  """__channel_bucket (uq utils.py:277-289): reshape to [-1, cout]; bucket = output channel."""
  cout = w.shape[-1]
  return f32(w).reshape(-1, cout), cout, 0


def uniform_quantize(x: np.ndarray, bits: int, mode: str = 'weight', use_buckets: bool = False,
                     bucket_type: str = 'channel', bucket_size: int = 256
                     ) -> Tuple[np.ndarray, Dict]:
  """__uniform_quantize (uq utils.py:163-199; copy for activations at nuq utils.py:245-282).

  Returns (fake-quantised tensor of x's shape, info dict with alpha/beta/bucket_num).
  Backward of the whole chain is the identity (gradient_override_map Round->Identity, :185), see
  `uniform_quantize_grad`.
  """
  x = f32(x)
  orig_shape = x.shape
  bucketed = use_buckets and mode == 'weight'
  bucket_num, padded_num = 0, 0
  if bucketed:
    if bucket_type == 'split':
      x, bucket_num, padded_num = split_bucket(x, bucket_size)
    elif bucket_type == 'channel':
      x, bucket_num, padded_num = channel_bucket(x)
    else:
      raise ValueError("Unrecognized bucket type, must be 'weight' or 'channel'.")
  axis = 0 if bucketed else None
  xn, alpha, beta = scale(x, axis)
  k = uq_k(bits)
  q = f32(np.rint(f32(xn * k)) / k)        # tf.round == round-half-to-even == np.rint
  y = inv_scale(q, alpha, beta)
  if bucketed:
    y = y.reshape(-1)
    if padded_num != 0:
      y = y[:-padded_num]
  y = y.reshape(orig_shape)
  return y, {'alpha': alpha, 'beta': beta, 'bucket_num': bucket_num, 'padded_num': padded_num,
             'bucket_storage_bits': bucket_num * 32 * 2}   # utils.py:299-306


def uniform_quantize_grad(g: np.ndarray) -> np.ndarray:
  """STE: d(out)/d(x) = alpha * (1/k) * 1 * k * (1/alpha) == 1  (SURVEY App. A.1.2)."""
  return f32(g)


ACTIVATIONS = {
    'Relu': lambda u: np.maximum(f32(u), np.float32(0)),
    'Relu6': lambda u: np.minimum(np.maximum(f32(u), np.float32(0)), np.float32(6)),
}


def activation_quantize(u: np.ndarray, bits: int, act: str = 'Relu') -> Tuple[np.ndarray, np.ndarray]:
  """insert_quant_op_for_activations (uq utils.py:51-79): t = act(u); per-tensor quantise t.

  Returns (quantised activation, the un-quantised activation t).  min/max span the WHOLE local
  batch tensor.  Always inserted, also at 32 bits (uq learner.py:38-39, 336).
  """
  t = ACTIVATIONS[act](u)
  y, _ = uniform_quantize(t, bits, mode='activation')
  return y, t


def activation_quantize_grad(g: np.ndarray, u: np.ndarray, act: str = 'Relu') -> np.ndarray:
  """STE through the quantiser, then the activation's own gradient (TF ReluGrad: g * (t > 0))."""
  u = f32(u)
  if act == 'Relu':
    return f32(g * (u > 0))
  if act == 'Relu6':
    return f32(g * ((u > 0) & (u < 6)))
  raise NotImplementedError(act)


# ---------------------------------------------------------------------------------------------
# percentile (third party: tf.contrib.distributions.percentile, interpolation='nearest')
# call sites: ws learner.py:285, nuq utils.py:363-364, ws pr_optimizer.py:270
# ---------------------------------------------------------------------------------------------

def percentile_nearest(x: np.ndarray, q, axis: Optional[int] = None) -> np.ndarray:
  """TF-1.x contrib percentile, 'nearest' interpolation.

  Sort DESCENDING (top_k), d = length, index = clip(int(round_half_even((d-1) * (1 - q/100))),
  0, d-1) computed in float64, return sorted[index].
  """
  x = f32(x)
  q = np.float64(q)
  if axis is None:
    y = x.reshape(-1)
  else:
    y = np.moveaxis(x, axis, -1)
  d = y.shape[-1]
  srt = -np.sort(-y, axis=-1, kind='stable')          # descending
  frac_at_q_or_above = np.float64(1.0) - q / np.float64(100.0)
  idx = int(np.clip(np.rint(np.float64(d - 1) * frac_at_q_or_above), 0, d - 1))
  return f32(srt[..., idx])


# ---------------------------------------------------------------------------------------------
# A.2  Non-uniform (codebook) fake quantisation   learners/nonuniform_quantization/utils.py
# ---------------------------------------------------------------------------------------------

def nuq_quantile_init(xn: np.ndarray, nb_clusters: int, use_buckets: bool) -> np.ndarray:
  """__quantile_init (nuq utils.py:349-366): c[i] = percentile(xn, (i+1)*100/(k+1)).

  xn is the normalised tensor ([bucket_size, bucket_num] if bucketed).  Returns [k] or
  [k, bucket_num].  q uses Python true division on int64 -> float64.
  """
  axis = 0 if use_buckets else None
  out = []
  for idx in range(nb_clusters):
    q = np.float64((idx + 1) * 100) / np.float64(nb_clusters + 1)
    out.append(percentile_nearest(xn, q, axis=axis))
  return f32(np.stack(out, axis=0))


def nuq_uniform_init(nb_clusters: int) -> np.ndarray:
  """__uniform_init without buckets (nuq utils.py:368-386): linspace(0, 1, k)."""
  return f32(np.linspace(0.0, 1.0, nb_clusters))


def nuq_assign(xn: np.ndarray, c: np.ndarray) -> np.ndarray:
  """argmin_j |xn - c_j| with ties -> lowest j (nuq utils.py:302 / :331).

  xn: any shape with c [k] (per-tensor), or xn [bucket_size, bucket_num] with c [k, bucket_num].
  """
  xn = f32(xn)
  c = f32(c)
  if c.ndim == 1:
    d = np.abs(xn[..., None] - c)                       # [..., k]
    return np.argmin(d, axis=-1).astype(np.int64)
  d = np.abs(xn[:, None, :] - c[None, :, :])            # [bucket_size, k, bucket_num]
  return np.argmin(d, axis=1).astype(np.int64)


def nuq_quantize(x: np.ndarray, bits: int, codebook: Optional[np.ndarray] = None,
                 use_buckets: bool = False, bucket_type: str = 'split', bucket_size: int = 256,
                 init_style: str = 'quantile') -> Tuple[np.ndarray, Dict]:
  """__nonuni_quantize / __bucket_quantize (nuq utils.py:168-243, 284-347), weights only.

  If `codebook` is None it is initialised from x (the value the 'clusters' variable takes when
  ops['cluster_init'] runs).  Returns (fake-quantised tensor, info with codebook/idx/alpha/beta).
  """
  x = f32(x)
  orig_shape = x.shape
  k = int(2 ** bits)
  padded_num, bucket_num = 0, 0
  if use_buckets:
    if bucket_type == 'split':
      xb, bucket_num, padded_num = split_bucket(x, bucket_size)
    elif bucket_type == 'channel':
      xb, bucket_num, padded_num = channel_bucket(x)
    else:
      raise ValueError(bucket_type)
  else:
    xb = x
  xn, alpha, beta = scale(xb, 0 if use_buckets else None)
  if codebook is None:
    if init_style == 'quantile':
      codebook = nuq_quantile_init(xn, k, use_buckets)
    elif init_style == 'uniform':
      if use_buckets:
        raise ValueError('bucketed uniform init is broken in the reference (SURVEY A.9-3)')
      codebook = nuq_uniform_init(k)
    else:
      raise ValueError('Unrecognized Initialization Mode.')
  codebook = f32(codebook)
  idx = nuq_assign(xn, codebook)
  if codebook.ndim == 1:
    gathered = codebook[idx]
  else:
    gathered = np.take_along_axis(codebook, idx, axis=0)       # c[idx[i,b], b]
  sgn = np.sign(f32(xn + np.float32(1e-6)))                     # == +1 (xn >= 0)
  qx = f32(gathered * sgn)
  y = inv_scale(qx, alpha, beta)
  if use_buckets:
    y = y.reshape(-1)
    if padded_num != 0:
      y = y[:-padded_num]
  y = y.reshape(orig_shape)
  return y, {'codebook': codebook, 'idx': idx, 'alpha': alpha, 'beta': beta, 'k': k,
             'bucket_num': bucket_num, 'padded_num': padded_num}


def nuq_backward(g: np.ndarray, info: Dict, use_buckets: bool = False, bucket_type: str = 'split',
                 bucket_size: int = 256) -> Tuple[np.ndarray, np.ndarray]:
  """Gradients of nuq_quantize under the override map {'Mul':'Add','Sign':'Identity'}
  (nuq utils.py:305-306, 345-346):

    gq = alpha * g ;  dL/dc_j = sum_{i: idx_i = j} gq_i (per bucket column) ;  dL/dx = gq/alpha = g.

  The padded tail of a split bucket receives zero upstream gradient (it is sliced away).
  """
  g = f32(g)
  k = info['k']
  idx = info['idx']
  alpha = info['alpha']
  if use_buckets:
    if bucket_type == 'split':
      gb, _, _ = split_bucket(g, bucket_size)
      if info['padded_num']:
        flat = gb.reshape(-1).copy()
        flat[-info['padded_num']:] = 0
        gb = flat.reshape(gb.shape)
    else:
      gb, _, _ = channel_bucket(g)
    gq = f32(gb * alpha)                                        # alpha [bucket_num] broadcasts
    dc = np.zeros((k, gb.shape[1]), dtype=np.float64)
    for b in range(gb.shape[1]):
      np.add.at(dc[:, b], idx[:, b], gq[:, b].astype(np.float64))
    return g, f32(dc)
  gq = f32(g * alpha)
  dc = np.zeros((k,), dtype=np.float64)
  np.add.at(dc, idx.reshape(-1), gq.reshape(-1).astype(np.float64))
  return g, f32(dc)


# ---------------------------------------------------------------------------------------------
# A.3  Weight sparsification     learners/weight_sparsification/{learner,utils,pr_optimizer}.py
# ---------------------------------------------------------------------------------------------

def get_maskable_var_names(names: Sequence[str]) -> List[str]:
  """get_maskable_vars (ws utils.py:19-39): 'kernel' | 'pointwise/weights' | 'Conv2d_1c_1x1/weights'."""
  out = []
  for n in names:
    if 'kernel' in n or 'pointwise/weights' in n or 'Conv2d_1c_1x1/weights' in n:
      out.append(n)
  return out


def pr_uniform(names: Sequence[str], prune_ratio: float) -> List[Tuple[str, float]]:
  """PROptimizer.__calc_uniform_prune_ratios (pr_optimizer.py:385-392)."""
  return [(n, prune_ratio) for n in names]


def pr_heurist(names: Sequence[str], nb_params: Sequence[int], prune_ratio: float
               ) -> List[Tuple[str, float]]:
  """__calc_heurist_prune_ratios (pr_optimizer.py:394-409): ratio_i = alpha * ln(n_i),
  alpha = prune_ratio * sum(n) / sum(n * ln n)."""
  n = np.array(nb_params, dtype=np.float64)
  alpha = prune_ratio * np.sum(n) / np.sum(n * np.log(n))
  return [(nm, float(alpha * np.log(ni))) for nm, ni in zip(names, n)]


def ws_prune_ratio_dyn(global_step: int, nb_iters_train: int, prune_ratio_fnl: float,
                       iter_ratio_beg: float = 0.1, iter_ratio_end: float = 0.5,
                       prune_ratio_exp: float = 3.0) -> np.float32:
  """__calc_prune_ratio_dyn (ws learner.py:296-312), float32 like the TF graph."""
  idx_iter_beg = int(nb_iters_train * iter_ratio_beg)
  idx_iter_end = int(nb_iters_train * iter_ratio_end)
  base = np.float32(np.float32(global_step - idx_iter_beg) / np.float32(idx_iter_end - idx_iter_beg))
  base = np.minimum(np.float32(1.0), np.maximum(np.float32(0.0), base))
  one = np.float32(1.0)
  return np.float32(np.float32(prune_ratio_fnl) *
                    (one - np.float32(np.power(one - base, np.float32(prune_ratio_exp)))))


def ws_refresh_steps(nb_iters_train: int, mask_update_step: int = 500,
                     iter_ratio_beg: float = 0.1, iter_ratio_end: float = 0.5) -> List[int]:
  """The idx_iter values (0-based) after which [prune_op, init_opt_op] run (ws learner.py:112-131)."""
  out = []
  last_mask_applied = False
  for idx_iter in range(nb_iters_train):
    if (idx_iter + 1) % mask_update_step == 0:
      iter_ratio = float(idx_iter + 1) / nb_iters_train
      if iter_ratio >= iter_ratio_beg:
        if iter_ratio <= iter_ratio_end:
          out.append(idx_iter)
        elif not last_mask_applied:
          last_mask_applied = True
          out.append(idx_iter)
  return out


def ws_mask_refresh(var: np.ndarray, bkup: np.ndarray, mask: np.ndarray, prune_ratio_dyn
                    ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.float32]:
  """One var's prune_op chain (ws learner.py:283-288).

    bkup <- where(mask > 0.5, var, bkup)
    thr  <- percentile(|bkup|, prune_ratio * 100)      (nearest rank, float32 ratio*100 -> double)
    mask <- float(|bkup| > thr)
    var  <- bkup * mask
  Returns (var, bkup, mask, thr).
  """
  var, bkup, mask = f32(var), f32(bkup), f32(mask)
  bkup = np.where(mask > 0.5, var, bkup).astype(np.float32)
  q = np.float32(np.float32(prune_ratio_dyn) * np.float32(100.0))
  thr = percentile_nearest(np.abs(bkup), np.float64(q))
  mask = (np.abs(bkup) > thr).astype(np.float32)
  var = f32(bkup * mask)
  return var, bkup, mask, np.float32(thr)


def calc_prune_ratio(vars_list: Sequence[np.ndarray]) -> np.float32:
  """calc_prune_ratio (ws learner.py:51-65): 1 - count_nonzero / size, float32."""
  nnz = sum(int(np.count_nonzero(v)) for v in vars_list)
  tot = sum(int(v.size) for v in vars_list)
  return np.float32(1.0) - np.float32(nnz) / np.float32(tot)


def masked_grad(g: np.ndarray, mask: np.ndarray) -> np.ndarray:
  """__calc_grads_pruned (ws learner.py:314-332; cp learner.py:406-419): g * mask."""
  return f32(f32(g) * f32(mask))


# ---------------------------------------------------------------------------------------------
# A.4  Channel pruning: fine-tune masks + host-side selector   learners/channel_pruning/*
# ---------------------------------------------------------------------------------------------

def cp_grad_mask(kernel_shape: Sequence[int], keep_in: np.ndarray, keep_out: np.ndarray) -> np.ndarray:
  """cp learner.py:406-419: mask = ones(HWIO); mask[:, :, ~keep_in, :] = 0; mask[:, :, :, ~keep_out] = 0."""
  mask = np.ones(tuple(kernel_shape), dtype=np.float32)
  mask[:, :, np.invert(np.asarray(keep_in, dtype=bool)), :] = 0
  mask[:, :, :, np.invert(np.asarray(keep_out, dtype=bool))] = 0
  return mask


def cp_select_by_magnitude(W2: np.ndarray, c_new: int) -> np.ndarray:
  """cp_lasso=False branch (channel_pruner.py:622-630): keep the c_new input channels with the
  largest sum |W| over (h, w, out)."""
  s = np.sum(np.abs(W2), axis=(0, 1, 3))
  order = np.argsort(-s, kind='stable')
  idxs = np.zeros(W2.shape[2], dtype=bool)
  idxs[order[:c_new]] = True
  return idxs


def cp_lasso_select(X: np.ndarray, Y: np.ndarray, W2: np.ndarray, c_new: int,
                    rng: np.random.RandomState, alpha: float = 1e-4, tolerance: float = 0.02,
                    max_probe: int = 200) -> Tuple[np.ndarray, np.ndarray]:
  """compute_pruned_kernel + prune_kernel's reshape (channel_pruner.py:456-577, 588-640) with the
  real scikit-learn solvers (cp_quadruple=False).

  X: [n, kh, kw, cin] sampled input patches, Y: [n, cout] target outputs, W2: [kh, kw, cin, cout].
  `samples = randint(0, n, min(400, n // 20))`; design[(s, o), c] = sum_hw X[s,h,w,c] W2[h,w,c,o];
  LassoLars(alpha, fit_intercept=False, max_iter=3000); double `right` until nnz < c_new, then
  bisect (first probe at the INITIAL alpha, alpha <- (left+right)/2 at the end of each round) until
  c_new - tol*cin/2 <= nnz <= c_new + tol*cin/2, widening a collapsed bracket; finally
  LinearRegression(fit_intercept=False) of Y on the kept channels -> newW2 [kh, kw, c_kept, cout].
  `max_probe` only bounds pathological inputs (the reference loops unboundedly).
  """
  from sklearn.linear_model import LassoLars, LinearRegression
  X = np.asarray(X)
  Y = np.asarray(Y)
  nb_samples = X.shape[0]
  c_in = X.shape[-1]
  c_out = W2.shape[-1]
  samples = rng.randint(0, nb_samples, min(400, nb_samples // 20))
  reshape_X = np.rollaxis(
      np.transpose(X, (0, 3, 1, 2)).reshape((nb_samples, c_in, -1))[samples], 1, 0)
  reshape_W2 = np.transpose(np.transpose(W2, (3, 2, 0, 1)).reshape((c_out, c_in, -1)), [1, 2, 0])
  product = np.matmul(reshape_X, reshape_W2).reshape((c_in, -1)).T
  reshape_Y = Y[samples].reshape(-1)
  solver = LassoLars(alpha=alpha, fit_intercept=False, max_iter=3000)

  def solve(a):
    solver.alpha = a
    solver.fit(product, reshape_Y)
    idxs_ = solver.coef_ != 0.
    return idxs_, int(np.sum(idxs_))

  if c_new == c_in:
    idxs = np.array([True] * c_new)
  else:
    left, right = 0, alpha
    lbound = c_new - tolerance * c_in / 2
    rbound = c_new + tolerance * c_in / 2
    probes = 0
    while True:
      _, tmp = solve(right)
      probes += 1
      if tmp < c_new or probes > max_probe:
        break
      right *= 2
    while True:
      if lbound < 0:
        lbound = 1
      idxs, tmp = solve(alpha)
      probes += 1
      if (lbound <= tmp <= rbound) or probes > max_probe:
        break
      elif abs(left - right) <= right * 0.1:
        if lbound > 1:
          lbound = lbound - 1
        if rbound < c_in:
          rbound = rbound + 1
        left = left / 1.2
        right = right * 1.2
      elif tmp > rbound:
        left = left + (alpha - left) / 2
      else:
        right = right - (right - alpha) / 2
      if alpha < 1e-10:
        break
      alpha = (left + right) / 2
  reg = LinearRegression(fit_intercept=False)
  reg.fit(X[:, :, :, idxs].reshape((nb_samples, -1)), Y)
  kh, kw = W2.shape[0], W2.shape[1]
  nb_kept = int(np.sum(idxs))
  newW2 = reg.coef_.reshape(-1, kh, kw, nb_kept)
  newW2 = np.transpose(newW2, (1, 2, 3, 0))
  return idxs, f32(newW2)


def cp_prune_w1(father_kernel: np.ndarray, father_bias: Optional[np.ndarray], idxs: np.ndarray):
  """prune_W1 (channel_pruner.py:665-694): zero the producer's pruned output channels (+bias)."""
  k = f32(father_kernel).copy()
  k[..., np.invert(idxs)] = 0
  b = None
  if father_bias is not None:
    b = f32(father_bias).copy()
    b[np.invert(idxs)] = 0
  return k, b


def cp_prune_w2(kernel: np.ndarray, idxs: np.ndarray, newW2: np.ndarray) -> np.ndarray:
  """prune_W2 (channel_pruner.py:696-725): kernel[:, :, idxs, :] = newW2; kernel[:, :, ~idxs, :] = 0."""
  k = f32(kernel).copy()
  k[:, :, idxs, :] = newW2
  k[:, :, np.invert(idxs), :] = 0
  return k


# ---------------------------------------------------------------------------------------------
# A.5 / A.6  Losses       learners/distillation_helper.py:86-103 ; nets/*.py calc_loss
# ---------------------------------------------------------------------------------------------

def log_softmax(z: np.ndarray) -> np.ndarray:
  z = f32(z)
  m = z.max(axis=1, keepdims=True)
  s = f32(z - m)
  lse = f32(np.log(np.sum(np.exp(s), axis=1, keepdims=True, dtype=np.float32)))
  return f32(s - lse)


def softmax(z: np.ndarray) -> np.ndarray:
  z = f32(z)
  m = z.max(axis=1, keepdims=True)
  e = np.exp(f32(z - m))
  return f32(e / np.sum(e, axis=1, keepdims=True, dtype=np.float32))


def softmax_cross_entropy(labels: np.ndarray, logits: np.ndarray) -> Tuple[np.float32, np.ndarray]:
  """tf.losses.softmax_cross_entropy(onehot_or_soft_labels, logits): per-example
  -sum_c labels*log_softmax(logits), reduction SUM_BY_NONZERO_WEIGHTS == mean over the batch.
  Returns (loss, dloss/dlogits) with dlogits = (softmax(logits)*sum_c(labels) - labels) / B.
  """
  labels, logits = f32(labels), f32(logits)
  B = logits.shape[0]
  lsm = log_softmax(logits)
  per_ex = -np.sum(f32(labels * lsm), axis=1, dtype=np.float32)
  loss = np.float32(np.sum(per_ex, dtype=np.float32) / np.float32(B))
  lab_sum = np.sum(labels, axis=1, keepdims=True, dtype=np.float32)
  dlogits = f32((softmax(logits) * lab_sum - labels) / np.float32(B))
  return loss, dlogits


def distill_loss(logits_pri: np.ndarray, logits_dst: np.ndarray, tempr: float = 4.0,
                 loss_w: float = 4.0) -> Tuple[np.float32, np.ndarray]:
  """DistillationHelper.calc_loss (distillation_helper.py:86-103).

    L = loss_w * mean_n CE(softmax(z_t / T), z_s / T)        (no T^2 factor)
    dL/dz_s = loss_w / (B * T) * (softmax(z_s / T) - softmax(z_t / T))
  """
  T = np.float32(tempr)
  logits_soft = f32(f32(logits_pri) / T)
  labels_soft = softmax(f32(f32(logits_dst) / T))
  ce, dsoft = softmax_cross_entropy(labels_soft, logits_soft)
  loss = np.float32(np.float32(loss_w) * ce)
  dz = f32(np.float32(loss_w) * dsoft / T)
  return loss, dz


def l2_loss(v: np.ndarray) -> np.float32:
  """tf.nn.l2_loss: sum(v**2) / 2."""
  v = f32(v)
  return np.float32(np.sum(v.astype(np.float64) ** 2) / 2.0)


def model_loss(labels: np.ndarray, logits: np.ndarray, l2_vars: Sequence[np.ndarray],
               loss_w_dcy: float) -> Tuple[np.float32, np.ndarray, List[np.ndarray]]:
  """ModelHelper.calc_loss (nets/resnet_at_ilsvrc12.py:129-141 and siblings):
  CE(onehot, logits) + loss_w_dcy * sum_v l2_loss(v).  Returns (loss, dlogits, [dL/dv = wd * v])."""
  ce, dlogits = softmax_cross_entropy(labels, logits)
  reg = np.float32(0)
  for v in l2_vars:
    reg = np.float32(reg + l2_loss(v))
  loss = np.float32(ce + np.float32(loss_w_dcy) * reg)
  dvars = [f32(np.float32(loss_w_dcy) * f32(v)) for v in l2_vars]
  return loss, dlogits, dvars


def in_top_k(outputs: np.ndarray, targets: np.ndarray, k: int) -> np.ndarray:
  """tf.nn.in_top_k: target's score is among the k largest; ties count in favour (a target is in
  the top k if fewer than k entries are STRICTLY greater)."""
  outputs = f32(outputs)
  tgt = outputs[np.arange(outputs.shape[0]), targets][:, None]
  return (np.sum(outputs > tgt, axis=1) < k)


def metrics_ilsvrc(labels: np.ndarray, outputs: np.ndarray) -> Dict[str, np.float32]:
  """nets/resnet_at_ilsvrc12.py:136-139 -- note 'accuracy' := acc_top5."""
  targets = np.argmax(labels, axis=1)
  top1 = np.float32(np.mean(in_top_k(outputs, targets, 1).astype(np.float32)))
  top5 = np.float32(np.mean(in_top_k(outputs, targets, 5).astype(np.float32)))
  return {'accuracy': top5, 'acc_top1': top1, 'acc_top5': top5}


def metrics_cifar(labels: np.ndarray, outputs: np.ndarray) -> Dict[str, np.float32]:
  """nets/resnet_at_cifar10.py:106-110."""
  acc = np.mean((np.argmax(labels, axis=1) == np.argmax(outputs, axis=1)).astype(np.float32))
  return {'accuracy': np.float32(acc)}


# ---------------------------------------------------------------------------------------------
# A.7  Optimisers and schedules  (third party: tf.train.*)
# ---------------------------------------------------------------------------------------------

def piecewise_constant(step: int, boundaries: Sequence[int], values: Sequence[float]) -> float:
  """tf.train.piecewise_constant: values[0] if step <= b[0]; values[i] if b[i-1] < step <= b[i];
  values[-1] if step > b[-1]."""
  if step <= boundaries[0]:
    return values[0]
  for i in range(1, len(boundaries)):
    if boundaries[i - 1] < step <= boundaries[i]:
      return values[i]
  return values[-1]


def beta_power(beta: float, t: int) -> np.float32:
  """beta^t as TF keeps it: a float32 variable multiplied by beta after every step."""
  b = np.float32(beta)
  p = np.float32(beta)
  for _ in range(t - 1):
    p = np.float32(p * b)
  return p


def adam_step(p, g, m, v, t: int, lr: float, beta1: float = 0.9, beta2: float = 0.999,
              eps: float = 1e-8):
  """tf.train.AdamOptimizer._apply_dense (t = 1 for the first step):
    lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    m <- m + (g - m) * (1 - beta1) ; v <- v + (g*g - v) * (1 - beta2) ; p <- p - lr_t * m / (sqrt(v) + eps)
  (the C++ kernel's update form; epsilon OUTSIDE the sqrt, bias correction folded into lr_t)."""
  p, g, m, v = f32(p), f32(g), f32(m), f32(v)
  b1, b2 = np.float32(beta1), np.float32(beta2)
  one = np.float32(1)
  b1p, b2p = beta_power(beta1, t), beta_power(beta2, t)
  lr_t = np.float32(np.float32(lr) * np.sqrt(one - b2p) / (one - b1p))
  m = f32(m + f32(f32(g - m) * (one - b1)))
  v = f32(v + f32(f32(f32(g * g) - v) * (one - b2)))
  p = f32(p - f32(f32(lr_t * m) / f32(np.sqrt(v) + np.float32(eps))))
  return p, m, v


def momentum_step(p, g, acc, lr: float, momentum: float = 0.9):
  """tf.train.MomentumOptimizer (use_nesterov=False): acc <- mu*acc + g ; p <- p - lr*acc."""
  p, g, acc = f32(p), f32(g), f32(acc)
  acc = f32(f32(np.float32(momentum) * acc) + g)
  p = f32(p - f32(np.float32(lr) * acc))
  return p, acc


def lrn_rate_piecewise(global_step: int, batch_size_total: int, idxs_epoch: Sequence[float],
                       decay_rates: Sequence[float], nb_smpls_train: int, lrn_rate_init: float,
                       batch_size_norm: float, nb_epochs_rat: float = 1.0) -> float:
  """setup_lrn_rate_piecewise_constant (utils/lrn_rate_utils.py:23-46)."""
  idxs = [e * nb_epochs_rat for e in idxs_epoch]
  init = lrn_rate_init * batch_size_total / batch_size_norm
  nb_batches_per_epoch = float(nb_smpls_train) / batch_size_total
  bnds = [int(nb_batches_per_epoch * e) for e in idxs]
  vals = [init * d for d in decay_rates]
  return piecewise_constant(global_step, bnds, vals)


def uq_setup_bnds_decay_rates(model_name: str, dataset_name: str, batch_size: int,
                              nb_smpls_train: int, lrn_rate_init: float, batch_size_norm: float,
                              quant_epochs: int = 60, enbl_multi_gpu: bool = False, mgw_size: int = 1,
                              enbl_warm_start: bool = False):
  """setup_bnds_decay_rates of the UQ learner (uq learner.py:50-70).  For model/dataset pairs the
  reference has no table for (e.g. lenet: UnboundLocalError) we fall back to the
  cifar_10/resnet table -- a documented deviation."""
  bs = batch_size if not enbl_multi_gpu else batch_size * mgw_size
  nb = int(nb_smpls_train / bs)
  n = int(mgw_size) if enbl_multi_gpu else 1
  init_lr = lrn_rate_init * batch_size * n / batch_size_norm if enbl_multi_gpu else lrn_rate_init
  if dataset_name == 'ilsvrc_12' and model_name.startswith('resnet'):
    bnds, decay = [nb * 5, nb * 20], [1e-4, 1e-5, 1e-6]
  elif dataset_name == 'ilsvrc_12' and model_name.startswith('mobilenet'):
    bnds, decay = [nb * 5, nb * 30], [1e-4, 1e-5, 1e-6]
  else:
    bnds, decay = [nb * 15, nb * 40], [1e-3, 1e-4, 1e-5]
  steps = nb * quant_epochs
  init_lr = init_lr if enbl_warm_start else lrn_rate_init
  return init_lr, bnds, decay, steps


def nuq_setup_bnds_decay_rates(model_name: str, dataset_name: str, batch_size: int,
                               nb_smpls_train: int, lrn_rate_init: float, batch_size_norm: float,
                               quant_epochs: int = 60, enbl_multi_gpu: bool = False, mgw_size: int = 1,
                               enbl_warm_start: bool = False):
  """setup_bnds_decay_rates of the NUQ learner (nuq learner.py:52-73)."""
  bs = batch_size if not enbl_multi_gpu else batch_size * mgw_size
  nb = int(nb_smpls_train / bs)
  n = int(mgw_size) if enbl_multi_gpu else 1
  init_lr = lrn_rate_init * batch_size * n / batch_size_norm if enbl_multi_gpu else lrn_rate_init
  if dataset_name == 'ilsvrc_12' and model_name.startswith('resnet'):
    bnds, decay = [nb * 5, nb * 20], [5e-4, 5e-5, 5e-6]
  elif dataset_name == 'ilsvrc_12' and model_name.startswith('mobilenet'):
    bnds, decay = [nb * 5, nb * 30], [1e-4, 1e-5, 1e-6]
  else:
    bnds, decay = [nb * 40, nb * 80], [1e-4, 1e-5, 1e-6]
  steps = nb * quant_epochs
  init_lr = init_lr if enbl_warm_start else lrn_rate_init
  return init_lr, bnds, decay, steps


# ---------------------------------------------------------------------------------------------
# Batch normalisation (third party: tf.layers.batch_normalization, fused=True)  resnet_model.py:55-62
# ---------------------------------------------------------------------------------------------

def batch_norm_train(x: np.ndarray, gamma, beta, moving_mean, moving_var, momentum: float,
                     eps: float):
  """Training-mode fused batch norm over NHWC (reduce over all but the last axis).

  y = gamma * (x - mean) / sqrt(var_biased + eps) + beta; the moving average is fed the UNBIASED
  variance.  Returns (y, new_moving_mean, new_moving_var, saved (mean, inv_std)).
  """
  x = f32(x)
  c = x.shape[-1]
  xr = x.reshape(-1, c).astype(np.float64)
  n = xr.shape[0]
  mean = xr.mean(axis=0)
  var = xr.var(axis=0)
  inv_std = 1.0 / np.sqrt(var + eps)
  y = (xr - mean) * inv_std * np.asarray(gamma, np.float64) + np.asarray(beta, np.float64)
  unbiased = var * (n / max(n - 1, 1))
  mm = np.asarray(moving_mean, np.float64) * momentum + mean * (1 - momentum)
  mv = np.asarray(moving_var, np.float64) * momentum + unbiased * (1 - momentum)
  return f32(y.reshape(x.shape)), f32(mm), f32(mv), (f32(mean), f32(inv_std))


def batch_norm_train_bwd(dy: np.ndarray, x: np.ndarray, gamma, mean, inv_std):
  """Backward of training-mode batch norm: returns (dx, dgamma, dbeta)."""
  c = x.shape[-1]
  dyr = f32(dy).reshape(-1, c).astype(np.float64)
  xr = f32(x).reshape(-1, c).astype(np.float64)
  n = xr.shape[0]
  xhat = (xr - np.asarray(mean, np.float64)) * np.asarray(inv_std, np.float64)
  dbeta = dyr.sum(axis=0)
  dgamma = (dyr * xhat).sum(axis=0)
  dx = (np.asarray(gamma, np.float64) * np.asarray(inv_std, np.float64)) * \
      (dyr - dbeta / n - xhat * dgamma / n)
  return f32(dx.reshape(x.shape)), f32(dgamma), f32(dbeta)


def batch_norm_eval(x, gamma, beta, moving_mean, moving_var, eps: float):
  x = f32(x).astype(np.float64)
  y = (x - np.asarray(moving_mean, np.float64)) / np.sqrt(np.asarray(moving_var, np.float64) + eps)
  return f32(y * np.asarray(gamma, np.float64) + np.asarray(beta, np.float64))


# ---------------------------------------------------------------------------------------------
# layout helpers used by the tests (reference HWIO  <->  product KRSC = [cout, kh, kw, cin])
# ---------------------------------------------------------------------------------------------

def hwio_to_krsc(w: np.ndarray) -> np.ndarray:
  if w.ndim == 2:                      # dense [in, out] -> [out, in]
    return np.ascontiguousarray(w.T)
  return np.ascontiguousarray(np.transpose(w, (3, 0, 1, 2)))


def krsc_to_hwio(w: np.ndarray) -> np.ndarray:
  if w.ndim == 2:
    return np.ascontiguousarray(w.T)
  return np.ascontiguousarray(np.transpose(w, (1, 2, 3, 0)))


# ---------------------------------------------------------------------------------------------------------------------------------
# 'chn-pruned-gpu': the stochastic proximal-gradient step (learners/channel_pruning_gpu/learner.py:379-383)
# ---------------------------------------------------------------------------------------------------------------------------------

def cpg_proximal_step(w_hwio: np.ndarray, g_hwio: np.ndarray, lrn_rate, prune_perctl) -> Tuple[np.ndarray, np.ndarray, np.float32]:
  """var_prnd_new = var_prnd - lrn_rate_pgd * grad                                          (:379)
  var_norm = tf.sqrt(tf.reduce_sum(tf.square(var_prnd_new), axis=[0, 1, 3], keepdims=True))  (:380)
  threshold = tf.contrib.distributions.percentile(var_norm, prune_perctl)                     (:381, 'nearest')
  shrk_vec = tf.maximum(1.0 - threshold / var_norm, 0.0)                                     (:382)
  prune_op = var_prnd.assign(var_prnd_new * shrk_vec)                                        (:383)
  One float32 rounding per TF op; pinned by tests/golden/reference_cpg.npz (the five statements executed over oracle/tf_stub.py).
  Returns (new kernel HWIO, var_norm [I], threshold).  A channel whose norm is exactly 0 under a
  zero threshold gives 0 / 0 in TF (nan * 0-weights = nan); the learner never reaches that state from finite weights with p > 0
  (the threshold is then a positive norm); this restatement maps it to 0 like the product (stated, not pinned)."""
  w = f32(w_hwio)
  g = f32(g_hwio)
  new = f32(w - f32(np.float32(lrn_rate) * g))
  norm = f32(np.sqrt(np.sum(np.square(new), axis=(0, 1, 3), keepdims=True, dtype=np.float32)))
  thr = np.float32(percentile_nearest(norm, np.float32(prune_perctl)))
  with np.errstate(divide='ignore', invalid='ignore'):
    shrk = f32(np.maximum(f32(np.float32(1.0) - f32(thr / norm)), np.float32(0.0)))
  shrk = np.where(np.isnan(shrk), np.float32(0.0), shrk).astype(np.float32)
  return f32(new * shrk), norm.reshape(-1), thr
