"""Size-independent properties of the CPU oracle (oracle/pf_oracle.py), checked with hypothesis on random shapes / values.
The fixtures of test_oracle_golden.py pin the oracle to the reference's code on a handful of tensors; these properties hold for
EVERY input and are the same ones the GPU parity tests use at full size (tests/test_kernels_gpu.py, parity_common.py), so a
disagreement between the two would show on the CPU first.  Runs without a GPU."""
import numpy as np
import pytest

pytest.importorskip('hypothesis')         # (a test-only dependency: the rest of the suite must not fail at collection without it)
from hypothesis import given, settings, strategies as st   # noqa: E402
from hypothesis.extra import numpy as hnp                  # noqa: E402

from oracle import pf_oracle as O

SETTINGS = dict(max_examples=60, deadline=None, derandomize=True)
finite = st.floats(min_value=-4.0, max_value=4.0, allow_nan=False, allow_infinity=False, width=32)


def tensors(min_dims=1, max_dims=4, min_side=1, max_side=7):
  return hnp.arrays(np.float32, hnp.array_shapes(min_dims=min_dims, max_dims=max_dims, min_side=min_side, max_side=max_side), elements=finite)


@settings(**SETTINGS)
@given(tensors(), st.sampled_from([1, 2, 3, 4, 8]))
def test_uniform_quantize_grid_range_and_order(w, bits):
  """The fake-quantised tensor has at most 2**bits distinct values per range, stays inside [min, max] of its input up to the
  rounding of the inverse scaling, and preserves the order of the inputs (a monotone map)."""
  q, info = O.uniform_quantize(w, bits)
  assert q.shape == w.shape and q.dtype == np.float32
  assert len(np.unique(q)) <= 2 ** bits
  lo, hi = float(w.min()), float(w.max())
  span = max(hi - lo, 1e-6)
  assert float(q.min()) >= lo - 1e-5 * span - 1e-6 and float(q.max()) <= hi + 1e-5 * span + 1e-6
  order = np.argsort(w.reshape(-1), kind='stable')
  assert np.all(np.diff(q.reshape(-1)[order]) >= 0)
  # the straight-through estimator: the gradient of the chain is the identity
  g = np.arange(w.size, dtype=np.float32).reshape(w.shape)
  assert np.array_equal(O.uniform_quantize_grad(g), g)


@settings(**SETTINGS)
@given(tensors(min_dims=2), st.sampled_from([2, 4, 8]))
def test_uniform_quantize_is_idempotent_at_8_bits_or_fewer(w, bits):
  """Quantising a quantised tensor again (same bit width, range re-derived from the quantised values: the extremes are grid
  points) changes no value by more than one float32 rounding of the inverse scaling."""
  q1, _ = O.uniform_quantize(w, bits)
  q2, _ = O.uniform_quantize(q1, bits)
  span = max(float(w.max() - w.min()), 1e-6)
  assert float(np.max(np.abs(q2 - q1))) <= 4e-7 * span + 1e-7 * float(np.max(np.abs(w))) + 1e-12


@settings(**SETTINGS)
@given(tensors(min_dims=2, max_dims=4, min_side=2, max_side=6), st.sampled_from([2, 4, 8]), st.sampled_from(['channel', 'split']),
       st.sampled_from([4, 16, 256]))
def test_bucketed_uniform_quantize_quantises_every_bucket_on_its_own(w, bits, btype, bsize):
  q, info = O.uniform_quantize(w, bits, 'weight', True, btype, bsize)
  assert q.shape == w.shape
  if btype == 'channel':
    cols = w.reshape(-1, w.shape[-1])
    qc = q.reshape(-1, w.shape[-1])
    assert info['bucket_num'] == w.shape[-1]
    for c in range(cols.shape[1]):
      ref, _ = O.uniform_quantize(cols[:, c], bits)
      assert np.array_equal(qc[:, c], ref), c
  else:
    # utils.py:247-275: the flat vector is padded with copies of its LAST element to a multiple of the bucket size and reshaped
    # [bucket_size, multiple]: bucket j is the STRIDED set flat[j], flat[multiple + j], ... -- one range per column
    flat = w.reshape(-1)
    n = flat.size
    nb = -(-n // bsize)
    assert info['bucket_num'] == nb and info['padded_num'] == nb * bsize - n
    padded = np.concatenate([flat, np.full(nb * bsize - n, flat[-1], np.float32)]).reshape(bsize, nb)
    ref = np.stack([O.uniform_quantize(padded[:, j], bits)[0] for j in range(nb)], axis=1).reshape(-1)[:n]
    assert np.array_equal(q.reshape(-1), ref)


@settings(**SETTINGS)
@given(hnp.arrays(np.float32, st.integers(1, 200), elements=finite), st.floats(0.0, 100.0))
def test_percentile_nearest_is_an_order_statistic(x, q):
  v = O.percentile_nearest(x, q)
  assert v in x
  assert O.percentile_nearest(x, 0.0) == x.min() and O.percentile_nearest(x, 100.0) == x.max()
  # monotone in q, and at least round-to-nearest((d - 1) * q / 100) elements lie at or below it
  assert O.percentile_nearest(x, min(q + 7.0, 100.0)) >= v
  d = x.size
  idx_from_top = int(np.clip(np.rint((d - 1) * (1.0 - q / 100.0)), 0, d - 1))
  assert v == np.sort(x)[::-1][idx_from_top]


@settings(**SETTINGS)
@given(hnp.arrays(np.float32, st.tuples(st.integers(1, 3), st.integers(1, 3), st.integers(2, 12), st.integers(1, 9)), elements=finite),
       st.floats(0.05, 0.95))
def test_ws_mask_refresh_keeps_the_largest_magnitudes(var, ratio):
  """After a refresh: the mask is binary, var == bkup * mask, every kept magnitude exceeds every pruned one, the number of pruned
  weights is what the nearest-rank threshold implies, and refreshing again with the same ratio changes nothing (the pruned
  weights are restored from the backup first: learner.py:283)."""
  bkup0 = np.zeros_like(var)
  mask0 = np.ones_like(var)
  v1, b1, m1, thr = O.ws_mask_refresh(var, bkup0, mask0, ratio)
  assert set(np.unique(m1)) <= {0.0, 1.0}
  assert np.array_equal(b1, var) and np.array_equal(v1, b1 * m1)
  kept, pruned = np.abs(b1[m1 > 0.5]), np.abs(b1[m1 < 0.5])
  if kept.size and pruned.size:
    assert kept.min() > pruned.max()
  assert np.all(pruned <= thr) and np.all(kept > thr)
  v2, b2, m2, thr2 = O.ws_mask_refresh(v1, b1, m1, ratio)
  assert np.array_equal(m2, m1) and np.array_equal(v2, v1) and np.array_equal(b2, b1) and thr2 == thr
  # masked gradient: exactly zero where the mask is
  g = np.ones_like(var)
  assert np.array_equal(O.masked_grad(g, m1), m1)


@settings(**SETTINGS)
@given(hnp.arrays(np.float32, st.tuples(st.integers(1, 3), st.integers(1, 3), st.integers(2, 10), st.integers(1, 8)), elements=finite),
       st.floats(0.0, 1.0), st.floats(1.0, 99.0))
def test_cpg_proximal_step_shrinks_whole_input_channels(w, lr, q):
  """chn-pruned-gpu: the threshold is one of the channel norms, channels at or below it become exactly zero, the others are scaled
  by one factor in (0, 1) each (a channel keeps its direction), and a zero gradient with q -> 0 prunes at most the weakest channel."""
  g = np.roll(w, 1, axis=2) * np.float32(0.5)
  new, norm, thr = O.cpg_proximal_step(w, g, lr, q)
  assert new.shape == w.shape and norm.shape == (w.shape[2],) and np.all(norm >= 0)
  assert thr in norm
  stepped = (w - np.float32(lr) * g).astype(np.float32)
  for c in range(w.shape[2]):
    if norm[c] <= thr:
      assert not new[:, :, c, :].any(), c
    else:
      f = np.float32(1.0) - thr / norm[c]
      assert 0.0 < f <= 1.0
      np.testing.assert_allclose(new[:, :, c, :], stepped[:, :, c, :] * f, rtol=2e-6, atol=1e-7)


@settings(**SETTINGS)
@given(tensors(min_dims=2, max_dims=2, min_side=2, max_side=9), st.sampled_from([1, 2, 4]))
def test_nuq_quantize_maps_every_weight_to_its_nearest_codeword(w, bits):
  q, info = O.nuq_quantize(w, bits)
  code = np.asarray(info['codebook']).reshape(-1)
  assert q.shape == w.shape and code.size == 2 ** bits
  assert len(np.unique(q)) <= 2 ** bits
  # de-normalised codewords: q takes only values alpha * c + beta; the assignment is the nearest codeword in normalised space
  xn, alpha, beta = O.scale(w, None)
  idx = O.nuq_assign(xn.reshape(-1), code)
  d = np.abs(xn.reshape(-1, 1) - code.reshape(1, -1))
  assert np.array_equal(d[np.arange(d.shape[0]), idx], d.min(axis=1))


def test_losses_are_what_their_closed_forms_say():
  """distill_loss of identical logits is the entropy of the softened teacher distribution (cross-entropy(p, p) = H(p)) times loss_w; the model loss's L2 term is loss_w_dcy * sum ||v||^2 / 2."""
  rng = np.random.RandomState(0)
  z = rng.randn(6, 10).astype(np.float32)
  p = O.softmax(z / np.float32(4.0))
  h = np.float32(-(p * O.log_softmax(z / np.float32(4.0))).sum(axis=1).mean())
  got, dz = O.distill_loss(z, z, tempr=4.0, loss_w=1.0)
  assert abs(float(got) - float(h)) <= 1e-6 * abs(float(h))
  assert float(np.max(np.abs(dz))) <= 1e-7                    # student == teacher: the gradient vanishes
  got4, dz4 = O.distill_loss(z + np.float32(1.0), z, tempr=4.0, loss_w=4.0)   # softmax is shift-invariant; the weight is a factor
  assert abs(float(got4) - 4.0 * float(h)) <= 4e-6 * abs(float(h)) and float(np.max(np.abs(dz4))) <= 1e-6
  labels = np.eye(10, dtype=np.float32)[rng.randint(0, 10, 6)]
  vs = [rng.randn(3, 4).astype(np.float32), rng.randn(5).astype(np.float32)]
  loss, _, _ = O.model_loss(labels, z, vs, 3e-3)
  ce, _ = O.softmax_cross_entropy(labels, z)
  l2 = sum(float(np.sum(v.astype(np.float64) ** 2)) / 2 for v in vs)
  assert abs(float(loss) - (float(ce) + 3e-3 * l2)) <= 1e-5 * abs(float(loss))
