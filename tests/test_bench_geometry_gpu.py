"""Every convolution of ResNet-v2-50 at the BENCHMARKED geometry (batch 256, 224x224 -> 56 / 28 / 14 / 7 feature maps; what
`bench.py` times, BASELINE configs[2]) through the C ABI, against float32 / float64 torch references built from the SAME bf16
operands (the largest kernel-level cases were 40 x 28^2 and 3 x 56^2 -- other tile counts, split counts and
statistics-group counts than the 802 816-row launches of the bench).

One test per layer geometry and pass, with the operands the step gives the kernel (`pocketflow_amd/graph.py`):
  forward        1x1: BN + ReLU + 8-bit fake-quant prologue, [residual], statistics epilogue        (_run_conv1x1)
                 3x3: plain operands, statistics epilogue                                           (_run_conv2d)
  backward-data  1x1: plain / BN-backward sums of the producer BN in the epilogue / strided row map (_FusedConv1x1.backward)
                 3x3: flipped kernel + BN-backward sums / parity classes for stride 2               (_Conv2dIgemm.backward)
  backward-filter 1x1 with the prologue recomputed on the fly; 3x3 stride 1 / 2                     (conv1x1_wrw / conv2d_wrw)
The dispatcher (tile, split count, kernel family) is the product's own: no override is set."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

B = 256

# (H = W of the INPUT, K = input channels, N = output channels, residual operand in the forward epilogue)
CONV1X1 = [(56, 64, 64, False), (56, 64, 256, False), (56, 64, 256, True), (56, 256, 64, False), (56, 256, 128, False),
           (28, 128, 512, True), (28, 512, 128, False), (28, 512, 256, False), (14, 256, 1024, True), (14, 1024, 256, False),
           (14, 1024, 512, False), (7, 512, 2048, True), (7, 2048, 512, False)]
# strided projection shortcuts (input H, K, N), stride 2
PROJ = [(56, 256, 512), (28, 512, 1024), (14, 1024, 2048)]
# 3x3 convolutions (input H, C = N, stride)
CONV3X3 = [(56, 64, 1), (56, 128, 2), (28, 128, 1), (28, 256, 2), (14, 256, 1), (14, 512, 2), (7, 512, 1)]


@pytest.fixture(scope='module')
def hip():
  from pocketflow_amd import hip as h
  return h


def _bf(x):
  return x.to(torch.bfloat16)


def _rand(g, *shape, scale=1.0):
  return _bf(torch.randn(*shape, device='cuda', generator=g) * scale)


def _close(got, ref, what, scale=None, frac_tol=0.0):
  """Equal up to one bf16 ulp of the reference (or of `scale`, the magnitude of the terms the result is a sum of)."""
  got, ref = got.float(), ref.float()
  err = (got - ref).abs()
  mag = ref.abs() if scale is None else torch.maximum(ref.abs(), scale.float().abs())
  tol = mag * 2 ** -7 + 2e-2 * float(ref.abs().mean() + 1e-6)
  bad = float((err > tol).float().mean())
  assert bad <= frac_tol, '%s: %.3e of the elements differ by more than a bf16 ulp (max err %.3e, mean |ref| %.3e)' % (
      what, bad, float(err.max()), float(ref.abs().mean()))


def _prologue(hip, g, X, K, bits=8):
  """(scale_shift, slot, Q): BN constants, the activation range as pf_bn_finalize leaves it, and the stand-alone kernel's
  q = fake_quant(relu(scale * x + shift)) (bit-exact against the oracle: tests/test_kernels_gpu.py)."""
  M = X.numel() // K
  ss = torch.stack([torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g)])
  slot = torch.empty(2, dtype=torch.int32, device='cuda')
  hip.minmax_slots_init(slot)
  Q = torch.empty_like(X)
  hip.bn_act_quant_apply(X, Q, M, K, ss, 'Relu', None, 8, False)          # relu(bn(x)), bf16
  hip.minmax_tensor(torch.relu(X.float().reshape(M, K) * ss[0] + ss[1]).contiguous(), slot)
  hip.bn_act_quant_apply(X, Q, M, K, ss, 'Relu', slot, bits, True)
  return ss, slot, Q


def _check_stats(partial, y, what):
  assert not torch.isnan(partial).any(), what
  y = y.double()
  s_ref, q_ref = y.sum(0), (y * y).sum(0)
  torch.testing.assert_close(partial[:, 0].double().sum(0), s_ref, rtol=1e-4, atol=1e-4 * float(y.abs().sum(0).max()))
  torch.testing.assert_close(partial[:, 1].double().sum(0), q_ref, rtol=1e-4, atol=1e-6 * float(q_ref.max()))
  assert torch.equal(partial[:, 2].min(0).values, y.min(0).values.float()), what
  assert torch.equal(partial[:, 3].max(0).values, y.max(0).values.float()), what


def _bn_sums(hip, dq, x, M, C, ss, mi):
  nblk = 64
  p = torch.empty(nblk * 2 * C, device='cuda')
  hip.bn_bwd_stats(dq, x, M, C, ss, mi, 'Relu', p, nblk)
  dgamma, dbeta = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
  hip.bn_bwd_finalize(p, nblk, C, dgamma, dbeta)
  return dgamma, dbeta


@pytest.mark.parametrize('H,K,N,residual', CONV1X1)
def test_conv1x1_forward_at_bench_geometry(hip, H, K, N, residual):
  M = B * H * H
  g = torch.Generator(device='cuda').manual_seed(H + K + N)
  X = _rand(g, M, K, scale=2.0)
  W = _rand(g, N, K, scale=K ** -0.5)
  R = _rand(g, M, N) if residual else None
  ss, slot, Q = _prologue(hip, g, X, K)
  G = hip.conv1x1_stats_groups(M, N, K, prologue=True)
  partial = torch.full((G, 4, N), float('nan'), device='cuda')
  Y = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
  hip.conv1x1_fwd(X, W, Y, M, N, K, R=R, scale_shift=ss, act='Relu', slot=slot, bits=8, partial=partial)
  acc = Q.float() @ W.float().t()
  ref = _bf(acc + R.float()) if residual else _bf(acc)
  del X, Q
  # (the folded-constant fake-quant of the prologue may differ from the stand-alone chain on exact rounding ties: enumerated in
  # tests/test_conv_gpu.py::test_prologue_fake_quant_equals_oracle_except_enumerated_ties)
  _close(Y, ref, 'conv1x1 fwd %dx%d %d->%d' % (H, H, K, N), scale=acc, frac_tol=2e-3)
  del acc, ref
  _check_stats(partial, Y, 'statistics epilogue')
  # no prologue, no residual (the teacher's forward runs the SAME kernels with inference-mode BN constants; the first
  # convolution of stage 1 reads the materialised max-pool output)
  Xq = _rand(g, M, K)
  G0 = hip.conv1x1_stats_groups(M, N, K)
  p0 = torch.full((G0, 4, N), float('nan'), device='cuda')
  hip.conv1x1_fwd(Xq, W, Y, M, N, K, partial=p0)
  _close(Y, _bf(Xq.float() @ W.float().t()), 'conv1x1 fwd plain')
  _check_stats(p0, Y, 'statistics epilogue (plain)')


@pytest.mark.parametrize('H,K,N,residual', CONV1X1)
def test_conv1x1_backward_at_bench_geometry(hip, H, K, N, residual):
  M = B * H * H
  g = torch.Generator(device='cuda').manual_seed(H + K + N + 1)
  X = _rand(g, M, K, scale=2.0)
  W = _rand(g, N, K, scale=K ** -0.5)
  dY = _rand(g, M, N, scale=0.1)
  ss, slot, Q = _prologue(hip, g, X, K)
  mi = torch.stack([torch.randn(K, device='cuda', generator=g) * 0.1, torch.rand(K, device='cuda', generator=g) + 0.5])
  Wt = W.t().contiguous()
  # backward-data with the producer BN's backward sums in the epilogue
  Gd = hip.conv1x1_stats_groups(M, K, N)
  pd = torch.full((Gd, 2, K), float('nan'), device='cuda')
  dQ = torch.full((M, K), float('nan'), device='cuda', dtype=torch.bfloat16)
  hip.conv1x1_bwd_data_bnstats(dY, Wt, dQ, X, ss, mi, 'Relu', pd, M, N, K)
  ref = dY.float() @ W.float()
  _close(dQ, _bf(ref), 'conv1x1 bwd-data %dx%d %d->%d' % (H, H, K, N))
  dgamma, dbeta = _bn_sums(hip, dQ, X, M, K, ss, mi)
  dg2, db2 = torch.empty(K, device='cuda'), torch.empty(K, device='cuda')
  hip.bn_bwd_finalize(pd, Gd, K, dg2, db2)
  torch.testing.assert_close(db2, dbeta, rtol=2e-4, atol=1e-4 * float(dbeta.abs().max()) + 1e-3)
  torch.testing.assert_close(dg2, dgamma, rtol=2e-4, atol=1e-4 * float(dgamma.abs().max()) + 1e-3)
  # plain, and with a joined gradient (second consumer of a two-consumer activation: the residual operand)
  dQ2 = torch.empty_like(dQ)
  hip.conv1x1_fwd(dY, Wt, dQ2, M, K, N)
  assert torch.equal(dQ2, dQ), 'plain and statistics backward-data launches differ'
  Rj = _rand(g, M, K, scale=0.1)
  hip.conv1x1_fwd(dY, Wt, dQ2, M, K, N, R=Rj)
  _close(dQ2, _bf(ref + Rj.float()), 'conv1x1 bwd-data + joined gradient', scale=ref)
  del ref, dQ, dQ2, Rj
  # backward-filter with the quantising prologue, float32 gradient buffer (the product's flat gradient buffer is float32)
  ws = torch.empty((hip.conv1x1_wrw_splits(M, N, K) + 32) * N * K, device='cuda')
  dW = torch.full((N, K), float('nan'), device='cuda')
  hip.conv1x1_wrw(dY, X, dW, ws, M, N, K, scale_shift=ss, act='Relu', slot=slot, bits=8)
  refw = (dY.double().t() @ Q.double())
  sc = float(refw.abs().max())
  # float32 accumulation over M = %d rows in fixed-order slabs; quantiser ties of the on-the-fly prologue (2e-3 of the elements,
  # one grid step each) add noise of the order sqrt(M * 2e-3) * step * |dy|
  err = float((dW.double() - refw).abs().max()) / sc
  assert err <= 2e-3, ('conv1x1 wrw %dx%d %d->%d' % (H, H, K, N), err)
  dW2 = torch.empty_like(dW)
  hip.conv1x1_wrw(dY, X, dW2, ws, M, N, K, scale_shift=ss, act='Relu', slot=slot, bits=8)
  assert torch.equal(dW, dW2)
  # without the prologue: exact operands on both sides -> float32 accumulation order only
  hip.conv1x1_wrw(dY, X, dW, ws, M, N, K)
  refw = dY.double().t() @ X.double()
  err = float((dW.double() - refw).abs().max() / refw.abs().max())
  assert err <= 2e-5, ('conv1x1 wrw plain', err)


@pytest.mark.parametrize('H,K,N', PROJ)
def test_projection_shortcut_at_bench_geometry(hip, H, K, N):
  s = 2
  Ho = H // s
  M = B * Ho * Ho
  g = torch.Generator(device='cuda').manual_seed(H + K + N + 2)
  X = _rand(g, B, H, H, K, scale=2.0)
  W = _rand(g, N, K, scale=K ** -0.5)
  ss, slot, Q = _prologue(hip, g, X, K)
  geom = (Ho, Ho, H, H, s)
  G = hip.conv1x1_stats_groups(M, N, K, prologue=True)
  partial = torch.full((G, 4, N), float('nan'), device='cuda')
  Y = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
  hip.conv1x1_fwd(X, W, Y, M, N, K, scale_shift=ss, act='Relu', slot=slot, bits=8, partial=partial, geom=geom)
  Qs = Q[:, ::s, ::s, :].reshape(M, K)
  acc = Qs.float() @ W.float().t()
  _close(Y, _bf(acc), 'projection fwd %d %d->%d' % (H, K, N), scale=acc, frac_tol=2e-3)
  _check_stats(partial, Y, 'projection statistics')
  del acc
  dY = _rand(g, M, N, scale=0.1)
  Wt = W.t().contiguous()
  dX = torch.zeros(B, H, H, K, device='cuda', dtype=torch.bfloat16)
  hip.conv1x1_fwd(dY, Wt, dX, M, K, N, geom=geom, ymap=True)
  refd = torch.zeros(B, H, H, K, device='cuda')
  refd[:, ::s, ::s, :] = (dY.float() @ W.float()).reshape(B, Ho, Ho, K)
  _close(dX, _bf(refd), 'projection bwd-data (row map)')
  del refd, dX
  ws = torch.empty((hip.conv1x1_wrw_splits(M, N, K) + 32) * N * K, device='cuda')
  dW = torch.full((N, K), float('nan'), device='cuda')
  hip.conv1x1_wrw(dY, X, dW, ws, M, N, K, scale_shift=ss, act='Relu', slot=slot, bits=8, geom=geom)
  refw = dY.double().t() @ Qs.double()
  err = float((dW.double() - refw).abs().max() / refw.abs().max())
  assert err <= 2e-3, ('projection wrw', err)


def _conv_ref(x, w, stride):
  return F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=stride, padding=1).permute(0, 2, 3, 1)


@pytest.mark.parametrize('H,C,stride', CONV3X3)
def test_conv3x3_at_bench_geometry(hip, H, C, stride):
  N = C
  Ho = (H + 2 - 3) // stride + 1
  M = B * Ho * Ho
  g = torch.Generator(device='cuda').manual_seed(H + C + stride)
  x = _rand(g, B, H, H, C)
  w = _rand(g, N, 3, 3, C, scale=(9 * C) ** -0.5)
  geom = (B, H, H, C, N, 3, 3, stride, 1, 1, Ho, Ho)
  G = hip.conv2d_stats_groups(M, N, geom=geom)
  partial = torch.full((G, 4, N), float('nan'), device='cuda')
  y = torch.full((B, Ho, Ho, N), float('nan'), device='cuda', dtype=torch.bfloat16)
  hip.conv2d_fwd(x, w, y, B, H, H, C, N, 3, 3, stride, 1, 1, Ho, Ho, partial=partial)
  ref = _conv_ref(x, w, stride)
  _close(y, _bf(ref), 'conv3x3 fwd %dx%d %d s%d' % (H, H, C, stride))
  del ref
  _check_stats(partial, y.reshape(M, N), 'conv3x3 statistics')
  y2 = torch.empty_like(y)
  p2 = torch.full_like(partial, float('nan'))
  hip.conv2d_fwd(x, w, y2, B, H, H, C, N, 3, 3, stride, 1, 1, Ho, Ho, partial=p2)
  assert torch.equal(y, y2) and torch.equal(partial, p2)
  del y2, p2
  # backward-data
  dy = _rand(g, B, Ho, Ho, N, scale=0.1)
  xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
  F.conv2d(xt, w.float().permute(0, 3, 1, 2), stride=stride, padding=1).backward(dy.float().permute(0, 3, 1, 2))
  refd = xt.grad.permute(0, 2, 3, 1)
  del xt
  wb = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()            # [C][3][3][N]
  dx = torch.full((B, H, H, C), float('nan'), device='cuda', dtype=torch.bfloat16)
  if stride == 1:
    Mi = B * H * H
    bnx = _rand(g, Mi, C)
    ss = torch.stack([torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g) * 0.3])
    mi = torch.stack([torch.randn(C, device='cuda', generator=g) * 0.1, torch.rand(C, device='cuda', generator=g) + 0.5])
    Gd = hip.conv2d_stats_groups(Mi, C, geom=(B, H, H, N, C, 3, 3, 1, 1, 1, H, H))
    pd = torch.full((Gd, 2, C), float('nan'), device='cuda')
    hip.conv2d_fwd(dy, wb, dx, B, H, H, N, C, 3, 3, 1, 1, 1, H, H, partial=pd, bn_x=bnx, bn_scale_shift=ss, bn_mean_invstd=mi,
                   bn_act='Relu')
    _close(dx, _bf(refd), 'conv3x3 bwd-data')
    dgamma, dbeta = _bn_sums(hip, dx.reshape(Mi, C), bnx, Mi, C, ss, mi)
    dg2, db2 = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    hip.bn_bwd_finalize(pd, Gd, C, dg2, db2)
    torch.testing.assert_close(db2, dbeta, rtol=2e-4, atol=1e-4 * float(dbeta.abs().max()) + 1e-3)
    torch.testing.assert_close(dg2, dgamma, rtol=2e-4, atol=1e-4 * float(dgamma.abs().max()) + 1e-3)
  else:
    hip.conv2d_bwd_data_strided(dy, wb, dx, B, H, H, C, N, 3, 3, stride, 1, 1, Ho, Ho)
    assert torch.isfinite(dx.float()).all()
    _close(dx, _bf(refd), 'conv3x3 strided bwd-data')
    # the step's form: bn2's backward sums in the class launches' epilogues
    Mi = B * H * H
    bnx = _rand(g, Mi, C)
    ss = torch.stack([torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g) * 0.3])
    mi = torch.stack([torch.randn(C, device='cuda', generator=g) * 0.1, torch.rand(C, device='cuda', generator=g) + 0.5])
    Gd = hip.conv2d_bwd_data_strided_stats_groups(B, H, H, C, stride)
    pd = torch.full((Gd, 2, C), float('nan'), device='cuda')
    dx2 = torch.full_like(dx, float('nan'))
    hip.conv2d_bwd_data_strided(dy, wb, dx2, B, H, H, C, N, 3, 3, stride, 1, 1, Ho, Ho, partial=pd, bn_x=bnx, bn_scale_shift=ss,
                                bn_mean_invstd=mi, bn_act='Relu')
    assert torch.equal(dx2, dx) and not torch.isnan(pd).any()
    dgamma, dbeta = _bn_sums(hip, dx.reshape(Mi, C), bnx, Mi, C, ss, mi)
    dg2, db2 = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    hip.bn_bwd_finalize(pd, Gd, C, dg2, db2)
    torch.testing.assert_close(db2, dbeta, rtol=2e-4, atol=1e-4 * float(dbeta.abs().max()) + 1e-3)
    torch.testing.assert_close(dg2, dgamma, rtol=2e-4, atol=1e-4 * float(dgamma.abs().max()) + 1e-3)
    del dx2
  del refd, dx
  # backward-filter (float32 gradient buffer), against float64
  S = hip.conv2d_wrw_splits(M, N, C, 9)
  assert S > 0
  ws = torch.empty((S + 32) * N * 9 * C, device='cuda')
  dw = torch.full((N, 3, 3, C), float('nan'), device='cuda')
  hip.conv2d_wrw(dy, x, dw, ws, B, H, H, C, N, 3, 3, stride, 1, 1, Ho, Ho)
  refw = torch.empty(N, 3, 3, C, device='cuda', dtype=torch.float64)
  xp = F.pad(x, (0, 0, 1, 1, 1, 1))
  dyd = dy.reshape(M, N).double()
  for r in range(3):
    for s in range(3):
      xs = xp[:, r:r + stride * Ho:stride, s:s + stride * Ho:stride, :].reshape(M, C).double()
      refw[:, r, s, :] = dyd.t() @ xs
  err = float((dw.double() - refw).abs().max() / refw.abs().max())
  assert err <= 2e-5, ('conv3x3 wrw %dx%d %d s%d' % (H, H, C, stride), err)
  dw2 = torch.empty_like(dw)
  hip.conv2d_wrw(dy, x, dw2, ws, B, H, H, C, N, 3, 3, stride, 1, 1, Ho, Ho)
  assert torch.equal(dw, dw2)


def test_stem_and_pool_at_bench_geometry(hip):
  """7x7/2 3 -> 64 on 224x224 and the 3x3/2 max-pool on 112x112, batch 256."""
  H = 224
  g = torch.Generator(device='cuda').manual_seed(3)
  x = _rand(g, B, H, H, 3, scale=60.0)
  w = _rand(g, 64, 7, 7, 3, scale=0.05)
  y = torch.full((B, H // 2, H // 2, 64), float('nan'), device='cuda', dtype=torch.bfloat16)
  hip.conv_stem_fwd(x, w, y, B, H, H)
  ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=2, padding=3).permute(0, 2, 3, 1)
  _close(y, _bf(ref), 'stem fwd')
  del ref
  dy = _rand(g, B, H // 2, H // 2, 64, scale=0.1)
  S = hip.conv_stem_wrw_slabs(B, H, H)
  ws = torch.empty((S + 32) * 64 * 147, device='cuda')
  dw = torch.full((64, 7, 7, 3), float('nan'), device='cuda')
  hip.conv_stem_wrw(dy, x, dw, ws, B, H, H)
  refw = torch.ops.aten.convolution_backward(dy.double().permute(0, 3, 1, 2), x.double().permute(0, 3, 1, 2),
                                             torch.zeros(64, 3, 7, 7, device='cuda', dtype=torch.float64), None, [2, 2], [3, 3],
                                             [1, 1], False, [0, 0], 1, [False, True, False])[1].permute(0, 2, 3, 1)
  err = float((dw.double() - refw).abs().max() / refw.abs().max())
  assert err <= 1e-4, err


@pytest.mark.parametrize('H,K,N', [(56, 256, 64), (56, 64, 64), (56, 256, 128), (28, 512, 128), (14, 1024, 256), (7, 2048, 512)])
def test_teacher_conv1_with_bn2_in_the_epilogue_at_bench_geometry(hip, H, K, N):
  """The distillation teacher's conv1 launches (bn1 prologue without a quantiser, bn2 + ReLU folded into the epilogue:
  pf_conv1x1_fwd_affine) at batch 256 == the plain launch followed by the stand-alone pass, bit for bit."""
  M = B * H * H
  g = torch.Generator(device='cuda').manual_seed(H + K + N + 5)
  X = _rand(g, M, K, scale=2.0)
  W = _rand(g, N, K, scale=K ** -0.5)
  ss = torch.stack([torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g)])
  oss = torch.stack([torch.rand(N, device='cuda', generator=g) + 0.5, torch.randn(N, device='cuda', generator=g)])
  Y0 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  hip.conv1x1_fwd(X, W, Y0, M, N, K, scale_shift=ss, act='Relu')
  ref = torch.empty_like(Y0)
  hip.bn_act_quant_apply(Y0, ref, M, N, oss, 'Relu', None, 8, False)
  Y1 = torch.full_like(Y0, float('nan'))
  hip.conv1x1_fwd(X, W, Y1, M, N, K, scale_shift=ss, act='Relu', out_scale_shift=oss, out_act='Relu')
  assert torch.equal(Y1, ref)


@pytest.mark.parametrize('H,C,stride', CONV3X3)
def test_teacher_conv2_with_bn3_in_the_epilogue_at_bench_geometry(hip, H, C, stride):
  Ho = (H + 2 - 3) // stride + 1
  g = torch.Generator(device='cuda').manual_seed(H + C + stride + 7)
  x = _rand(g, B, H, H, C)
  w = _rand(g, C, 3, 3, C, scale=(9 * C) ** -0.5)
  oss = torch.stack([torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g) * 0.5])
  y0 = torch.empty(B, Ho, Ho, C, device='cuda', dtype=torch.bfloat16)
  hip.conv2d_fwd(x, w, y0, B, H, H, C, C, 3, 3, stride, 1, 1, Ho, Ho)
  ref = torch.empty_like(y0)
  hip.bn_act_quant_apply(y0, ref, B * Ho * Ho, C, oss, 'Relu', None, 8, False)
  y1 = torch.full_like(y0, float('nan'))
  hip.conv2d_fwd(x, w, y1, B, H, H, C, C, 3, 3, stride, 1, 1, Ho, Ho, out_scale_shift=oss, out_act='Relu')
  assert torch.equal(y1, ref)
