"""Per-layer A/B of the fused 1x1 convolutions on the ResNet-50 B=256 shapes: resident-kernel variant (pf_conv_stream.hip)
vs the tiled variant (pf_conv.hip), forward (prologue + residual + statistics), backward-data (+ BN-backward statistics)
and plain; every timed call is also checked against a float32 torch reference once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip


def timeit(fn, n=10):
  for _ in range(2): fn()
  torch.cuda.synchronize()
  a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n * 1e3   # us


B = int(os.environ.get('B', 256))
shapes = [(56, 64, 64, 0), (56, 64, 256, 1), (56, 256, 64, 0), (56, 256, 128, 0), (28, 128, 512, 1), (28, 512, 128, 0),
          (28, 512, 256, 0), (14, 256, 1024, 1), (14, 1024, 256, 0), (14, 1024, 512, 0), (7, 512, 2048, 1), (7, 2048, 512, 0)]
if os.environ.get('SHAPES'):
  shapes = [tuple(int(v) for v in s.split(',')) for s in os.environ['SHAPES'].split(';')]
print('%-16s | %-34s | %-24s | %-20s | floor(6.3TB/s)' % ('HW,K,N,res', 'fwd fused: stream  tiled  (err)', 'bwd-data+stats: str tiled', 'plain: str tiled'))
for hw, K, N, res in shapes:
  M = B * hw * hw
  g = torch.Generator(device='cuda').manual_seed(hw + K + N)
  X = (torch.randn(M, K, device='cuda', generator=g)).bfloat16()
  W = (torch.randn(N, K, device='cuda', generator=g) * 0.05).bfloat16()
  R = torch.randn(M, N, device='cuda', generator=g).bfloat16() if res else None
  ss = torch.stack([torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g)])
  slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
  hip.minmax_tensor(torch.relu(X.float() * ss[0] + ss[1]).contiguous(), slot)
  Y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  out = {}
  for mode in ('1', '0'):
    os.environ['PF_CONV_STREAM'] = mode
    hip.tuning_reload()          # the library reads its switches once
    G = hip.conv1x1_stats_groups(M, N, K, prologue=True)
    partial = torch.empty(G, 4, N, device='cuda')
    f = lambda: hip.conv1x1_fwd(X, W, Y, M, N, K, R=R, scale_shift=ss, act='Relu', slot=slot, bits=8, partial=partial)
    t_f = timeit(f)
    # check (sampled rows)
    Q = torch.empty(4096, K, device='cuda', dtype=torch.bfloat16)
    hip.bn_act_quant_apply(X[:4096].contiguous(), Q, 4096, K, ss, 'Relu', slot, 8, True)
    ref = (Q.float() @ W.float().t()).bfloat16().float()
    if res: ref = (ref + R[:4096].float()).bfloat16().float()
    err = float(((Y[:4096].float() - ref).abs() > (ref.abs() * 2 ** -6 + 2e-2)).float().mean())
    ssum = partial[:, 0].sum(0); sref = Y.float().sum(0)
    serr = float(((ssum - sref).abs() / (sref.abs() + 1.0 + 1e-3 * Y.float().abs().sum(0))).max())
    # backward data with BN-backward statistics: dQ[M][K] = dY[M][N] W[N][K]
    dY = torch.randn(M, N, device='cuda', generator=g).bfloat16()
    Wt = W.t().contiguous()
    dQ = torch.empty(M, K, device='cuda', dtype=torch.bfloat16)
    mi = torch.stack([torch.randn(K, device='cuda', generator=g) * 0.1, torch.rand(K, device='cuda', generator=g) + 0.5])
    G2 = hip.conv1x1_stats_groups(M, K, N)
    p2 = torch.empty(G2, 2, K, device='cuda')
    t_b = timeit(lambda: hip.conv1x1_bwd_data_bnstats(dY, Wt, dQ, X, ss, mi, 'Relu', p2, M, N, K))
    refb = (dY[:4096].float() @ W.float()).bfloat16().float()
    errb = float(((dQ[:4096].float() - refb).abs() > (refb.abs() * 2 ** -6 + 2e-2)).float().mean())
    t_p = timeit(lambda: hip.conv1x1_fwd(X, W, Y, M, N, K))
    out[mode] = (t_f, err, serr, t_b, errb, t_p)
    del dY, dQ
  floor = (M * K + (2 if res else 1) * M * N) * 2 / 6.3e12 * 1e6
  s, t = out['1'], out['0']
  print('%-16s | %7.0f %7.0f (%.0e %.0e | %.0e) | %7.0f %7.0f (%.0e) | %7.0f %7.0f | %6.0f' % (
      '%d,%d,%d,%d' % (hw, K, N, res), s[0], t[0], s[1], s[2], t[1], s[3], t[3], s[4], s[5], t[5], floor))
