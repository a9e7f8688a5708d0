"""Hand-derivable known answers for the oracle: the gradient (straight-through) rules, which the
forward-only reference fixtures of test_oracle_golden.py cannot pin, and a few closed forms.
Every expected value below is written out by hand (small integers / binary fractions are exact in
float32); nothing is produced by the code under test.  Runs without a GPU."""
import numpy as np

from oracle import pf_oracle as O


def test_round_half_even_ties_one_bit():
  # bits = 1 -> k = 1; x = [0, 1, 2]: alpha = 2 (+1e-10 vanishes in float32), x_hat = [0, .5, 1]
  # round_half_even(.5) = 0  => q = [0, 0, 1] => out = alpha*q + beta = [0, 0, 2]
  y, info = O.uniform_quantize(np.array([0., 1., 2.], np.float32), 1)
  assert y.tolist() == [0.0, 0.0, 2.0]
  assert float(info['alpha']) == 2.0 and float(info['beta']) == 0.0
  # x = [0, 3, 4]: x_hat = [0, .75, 1] -> [0, 1, 1] -> [0, 4, 4]
  y, _ = O.uniform_quantize(np.array([0., 3., 4.], np.float32), 1)
  assert y.tolist() == [0.0, 4.0, 4.0]


def test_constant_tensor_collapses_to_beta():
  # max == min: alpha = 1e-10, x_hat = 0 everywhere, out = beta (SURVEY 8c KAT 2)
  y, info = O.uniform_quantize(np.full((2, 3), 0.25, np.float32), 8)
  assert np.all(y == np.float32(0.25)) and np.float32(info['alpha']) == np.float32(1e-10)


def test_ste_rules():
  g = np.array([1., -2., 3., -4., 5.], np.float32)
  assert np.array_equal(O.uniform_quantize_grad(g), g)                       # Round -> Identity
  u = np.array([-1., 0., 2., 6., 7.], np.float32)
  assert O.activation_quantize_grad(g, u, 'Relu').tolist() == [0., 0., 3., -4., 5.]     # g * (u > 0)
  assert O.activation_quantize_grad(g, u, 'Relu6').tolist() == [0., 0., 3., 0., 0.]     # g * (0 < u < 6)


def test_nuq_argmin_tie_and_codebook_scatter():
  # x = [0, 1, 2, 4] -> alpha = 4, x_hat = [0, .25, .5, 1]; codebook [0, .5, 1] (k would be 2**bits; the
  # assignment itself only needs the array).  x_hat = .25 ties between c0 and c1 -> LOWEST index 0.
  xn = np.array([0., .25, .5, 1.], np.float32)
  c = np.array([0., .5, 1.], np.float32)
  assert O.nuq_assign(xn, c).tolist() == [0, 0, 1, 2]
  # gradient under {'Mul':'Add','Sign':'Identity'}: dL/dx = g ; dL/dc_j = sum_{idx_i = j} alpha * g_i
  info = {'k': 3, 'idx': np.array([0, 0, 1, 2]), 'alpha': np.float32(4.0), 'padded_num': 0}
  g = np.array([1., 2., -3., .5], np.float32)
  gw, dc = O.nuq_backward(g, info)
  assert gw.tolist() == g.tolist()
  assert dc.tolist() == [12.0, -12.0, 2.0]


def test_cp_gradient_mask():
  keep_in = np.array([True, False, True])
  keep_out = np.array([True, True, False, True])
  m = O.cp_grad_mask((1, 1, 3, 4), keep_in, keep_out)
  assert m[0, 0].tolist() == [[1, 1, 0, 1], [0, 0, 0, 0], [1, 1, 0, 1]]
  assert O.masked_grad(np.full((1, 1, 3, 4), 2.0, np.float32), m).sum() == 12.0


def test_momentum_exact():
  # acc <- .5*.25 + .5 = .625 ; p <- 1 - .5*.625 = .6875   (all exact binary fractions)
  p, acc = O.momentum_step(np.float32(1.0), np.float32(0.5), np.float32(0.25), 0.5, 0.5)
  assert float(p) == 0.6875 and float(acc) == 0.625


def test_adam_first_steps_closed_form():
  # t = 1, m = v = 0: m1 = (1-b1) g, v1 = (1-b2) g^2, lr_1 = lr sqrt(1-b2)/(1-b1)
  # => p1 = p0 - lr * g / (|g| + eps / sqrt(1-b2));   epsilon sits OUTSIDE the sqrt
  g = np.array([0.5, -2.0, 1e-3], np.float64)
  p0 = np.array([1.0, 1.0, 1.0], np.float64)
  lr, b1, b2, eps = 1e-2, 0.9, 0.999, 1e-8
  p1, m1, v1 = O.adam_step(p0, g, np.zeros(3), np.zeros(3), 1, lr)
  ref = p0 - lr * g / (np.abs(g) + eps / np.sqrt(1 - b2))
  assert np.max(np.abs(p1 - ref)) <= 2e-7
  assert np.max(np.abs(m1 - (1 - b1) * g)) <= 1e-7 and np.max(np.abs(v1 - (1 - b2) * g * g) / (g * g)) <= 1e-6
  # second step against the float64 recurrence
  g2 = np.array([-0.25, 1.0, 2e-3], np.float64)
  p2, m2, v2 = O.adam_step(p1, g2, m1, v1, 2, lr)
  m_ref = b1 * (1 - b1) * g + (1 - b1) * g2
  v_ref = b2 * (1 - b2) * g * g + (1 - b2) * g2 * g2
  lr_t = lr * np.sqrt(1 - b2 ** 2) / (1 - b1 ** 2)
  assert np.max(np.abs(p2 - (ref - lr_t * m_ref / (np.sqrt(v_ref) + eps)))) <= 5e-7


def test_softmax_ce_closed_forms():
  C = 8
  labels = np.eye(C, dtype=np.float32)[[1, 5]]
  loss, dz = O.softmax_cross_entropy(labels, np.zeros((2, C), np.float32))
  assert abs(float(loss) - np.log(C)) <= 1e-6                             # uniform logits: log C
  assert np.allclose(dz, (np.full((2, C), 1.0 / C) - labels) / 2, atol=1e-7)   # (softmax - onehot) / B
  # distillation with identical teacher and student logits: gradient vanishes, loss = w * H(p_T)
  z = np.array([[0., 1., 2., 3.]], np.float32)
  loss, dz = O.distill_loss(z, z, 4.0, 4.0)
  p = np.exp(z / 4) / np.exp(z / 4).sum()
  assert abs(float(loss) - 4.0 * float(-(p * np.log(p)).sum())) <= 1e-5 and np.max(np.abs(dz)) <= 1e-7


def test_piecewise_constant_boundaries_inclusive_left_interval():
  b, v = [10, 20], [1.0, 0.1, 0.01]
  assert [O.piecewise_constant(s, b, v) for s in (0, 10, 11, 20, 21)] == [1.0, 1.0, 0.1, 0.1, 0.01]


def test_in_top_k_ties_count_in_favour():
  out = np.array([[0.1, 0.5, 0.5, 0.2]], np.float32)
  assert O.in_top_k(out, np.array([2]), 1).tolist() == [True]            # nothing is STRICTLY greater
  assert O.in_top_k(out, np.array([3]), 2).tolist() == [False]


def test_batch_norm_train_statistics():
  x = np.array([[[[1., 10.]], [[3., 30.]]]], np.float32)                  # N=1,H=2,W=1,C=2
  y, mm, mv, (mean, inv_std) = O.batch_norm_train(x, [1, 1], [0, 0], [0, 0], [1, 1], 0.5, 0.0)
  assert mean.tolist() == [2.0, 20.0]
  assert np.allclose(y.reshape(2, 2), [[-1, -1], [1, 1]])
  assert mm.tolist() == [1.0, 10.0]                                      # .5*0 + .5*mean
  assert np.allclose(mv, [0.5 * 1 + 0.5 * 2.0, 0.5 * 1 + 0.5 * 200.0])   # UNBIASED variance feeds the average
