"""Product implicit-GEMM kernels against the -DPF_IG_SGB build of the same file (tools/gpu/build_ablate.sh -> libig_sgb.so), in which
the order of fragment reads and MFMAs inside a k-step is prescribed with __builtin_amdgcn_sched_group_barrier: all eight fragment reads
of a half step first (prologue kernels), or -- kernels without the in-LDS pass -- the second half's fragments requested a group of
MFMAs ahead.  hipcc on its own keeps one kernel-fragment register and exposes an LDS round trip per four MFMAs (ISA: DESIGN.md 4.1).
Same arithmetic in the same order: the outputs must be BIT-identical.  Layers: the 3x3 forward convolutions, the deep plain 1x1 GEMMs
(backward-data of stages 3-4) and the prologue 1x1 layers of the roofline region (statistics on, conv3 with its residual).
hipGraph-replay timing, us."""
import ctypes, os, subprocess, sys
from ctypes import c_int, c_void_p, c_float
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import gpu_time_us as timeit

here = os.path.dirname(os.path.abspath(__file__))
sgb_path = os.path.join(here, '_build', 'libig_sgb.so')
sgb = ctypes.CDLL(sgb_path)
prod_path = os.path.join(os.path.dirname(os.path.dirname(here)), 'pocketflow_amd', 'csrc', 'libpocketflow_hip.so')


def mangled(path, name):
  return [l.split()[-1] for l in subprocess.run(['nm', '-D', path], capture_output=True, text=True).stdout.splitlines()
          if name in l and ' T ' in l][0]


libs = {'product': hip._lib, 'sgb': sgb}
conv1x1 = {'product': getattr(hip._lib, mangled(prod_path, 'pf_igemm_conv1x1')), 'sgb': getattr(sgb, mangled(sgb_path, 'pf_igemm_conv1x1'))}
B = int(os.environ.get('B', 256))
p = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
stream = lambda: c_void_p(torch.cuda.current_stream().cuda_stream)
print('%-34s | %9s %9s | %6s | %s' % ('layer', 'product', 'sgb', 'ratio', 'bit-identical'))

# 1. R x S forward and plain 1x1 (pf_conv2d_fwd: no prologue), with the consumer BN's statistics
for H, C, N, k in [(56, 64, 64, 3), (28, 128, 128, 3), (14, 256, 256, 3), (7, 512, 512, 3), (28, 512, 128, 1), (14, 1024, 256, 1), (14, 256, 1024, 1), (7, 2048, 512, 1)]:
  g = torch.Generator(device='cuda').manual_seed(H + C + N)
  x = torch.randn(B, H, H, C, device='cuda', generator=g).bfloat16()
  w = (torch.randn(N, k, k, C, device='cuda', generator=g) * 0.05).bfloat16()
  pad = (k - 1) // 2
  z = hip.zero_page(x.device)
  G = hip._lib.pf_conv2d_stats_groups(c_int(B * H * H), c_int(N))
  out, ts = {}, {}
  for name, lib in libs.items():
    y = torch.empty(B, H, H, N, device='cuda', dtype=torch.bfloat16)
    part = torch.zeros(G, 4, N, device='cuda')
    def call():
      assert lib.pf_conv2d_fwd(p(x), p(w), p(y), p(z), p(None), p(part), p(None), p(None), p(None), c_int(0), c_int(B), c_int(H), c_int(H),
                               c_int(C), c_int(N), c_int(k), c_int(k), c_int(1), c_int(pad), c_int(pad), c_int(H), c_int(H), stream()) == 0
    call()
    torch.cuda.synchronize()
    out[name] = (y, part.clone())
    ts[name] = timeit(call)
  same = torch.equal(out['product'][0], out['sgb'][0]) and torch.equal(out['product'][1], out['sgb'][1])
  print('%-34s | %9.1f %9.1f | %6.3f | %s' % ('conv %dx%d %d^2 %d->%d' % (k, k, H, C, N), ts['product'], ts['sgb'], ts['sgb'] / ts['product'], same))
  assert same

# 2. prologue 1x1 (BN + ReLU + 8-bit fake-quant of the input, statistics of the output, conv3 with its residual)
for H, K, N, res in [(28, 512, 256, 0), (14, 1024, 256, 0), (14, 256, 1024, 1), (14, 1024, 512, 0), (7, 2048, 512, 0), (7, 512, 2048, 1), (28, 512, 128, 0)]:
  M = B * H * H
  g = torch.Generator(device='cuda').manual_seed(H + K + N)
  X = torch.randn(M, K, device='cuda', generator=g).bfloat16()
  W = (torch.randn(N, K, device='cuda', generator=g) * 0.05).bfloat16()
  R = torch.randn(M, N, device='cuda', generator=g).bfloat16() if res else None
  ss = torch.stack([torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g)])
  slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
  hip.minmax_tensor(torch.relu(X[:65536].float() * ss[0] + ss[1]).contiguous(), slot)
  G = hip.conv1x1_stats_groups(M, N, K, prologue=True)
  out, ts = {}, {}
  for name, fn in conv1x1.items():
    Y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    part = torch.zeros(G, 4, N, device='cuda')
    def call():
      assert fn(p(X), p(W), p(Y), p(R), p(part), c_void_p(0), c_void_p(0), c_void_p(0), c_float(0), c_float(0), p(ss), p(slot),
                c_float(255.0), c_float(0.0), c_float(float('inf')), c_int(M), c_int(N), c_int(K), c_int(0), c_int(0), c_int(0), c_int(0), c_int(1),
                stream()) == 0
    call()
    torch.cuda.synchronize()
    out[name] = (Y, part.clone())
    ts[name] = timeit(call)
  same = torch.equal(out['product'][0], out['sgb'][0]) and torch.equal(out['product'][1], out['sgb'][1])
  print('%-34s | %9.1f %9.1f | %6.3f | %s' % ('prologue 1x1 %d^2 %d->%d%s' % (H, K, N, ' +res' if res else ''), ts['product'], ts['sgb'], ts['sgb'] / ts['product'], same))
  assert same
