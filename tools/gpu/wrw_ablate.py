"""Where does the time of the backward-filter kernel k_wrw2 go?  pf_wrw2_launch (the kernel WITHOUT the cross-split reduction)
of the product library next to ablation builds of pf_wrw.hip (tools/gpu/build_ablate.sh):
  full   the product kernel          nomma  no fragment reads / MFMAs (LDS-DMA + barriers + slab stores)
  nodma  no LDS-DMA                  nost   no slab stores         dmaonly  neither MFMAs nor slab stores
plus the reduction launch alone, per ResNet-50 layer at B = 256.  Slab bytes = splits x N x taps*C x 4 (written by the kernel,
read again by the reduction)."""
import ctypes, os, sys
from ctypes import c_int, c_void_p, c_int64
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip

here = os.path.dirname(os.path.abspath(__file__))
prod = ctypes.CDLL(hip.lib_path())      # NOT RTLD_GLOBAL: the ablation builds' template kernels would bind to the product's
libs = {'full': prod}
for n, name in ((1, 'nomma'), (2, 'nodma'), (4, 'nost'), (6, 'mmaonly')):
  libs[name] = ctypes.CDLL(os.path.join(here, '_build', 'libwrw_ablate%d.so' % n))
LAUNCH = '_Z14pf_wrw2_launchPKvS0_PfPKfiPKjiiiiiiiiiiiiiilP12ihipStream_t'
SPLITS = '_Z14pf_wrw2_splitsiiii'


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import gpu_time_us as timeit   # hipGraph replay: no host launch overhead in the numbers


B = int(os.environ.get('B', 256))
# (H, C, N, k, prologue)
shapes = [(14, 1024, 256, 1, 1), (14, 256, 1024, 1, 1), (7, 2048, 512, 1, 1), (28, 512, 128, 1, 1), (56, 256, 64, 1, 1), (56, 64, 256, 1, 1),
          (14, 256, 256, 3, 0), (28, 128, 128, 3, 0), (7, 512, 512, 3, 0)]
print('%-16s S    | %7s %7s %7s %7s %8s | %9s | %8s %6s | %s' % ('H,C,N,k,pro', 'full', 'nomma', 'nodma', 'nost', 'mmaonly', 'reduce us', 'slab MB', 'in MB', 'TF(full+reduce)'))
for H, C, N, k, pro in shapes:
  g = torch.Generator(device='cuda').manual_seed(H + C + N)
  M = B * H * H
  x = torch.randn(M, C, device='cuda', generator=g).bfloat16()
  dy = (torch.randn(M, N, device='cuda', generator=g) * 0.1).bfloat16()
  ss = torch.stack([torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g)])
  slot = torch.empty(2, dtype=torch.int32, device='cuda'); hip.minmax_slots_init(slot)
  hip.minmax_tensor(torch.relu(x[:65536].float() * ss[0] + ss[1]).contiguous(), slot)
  taps = k * k
  S = getattr(prod, SPLITS)(c_int(M), c_int(N), c_int(C), c_int(taps))
  ws = torch.empty((S + 32) * N * taps * C, device='cuda')
  dw = torch.empty(N, taps * C, device='cuda')
  pad = (k - 1) // 2
  p = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
  st = c_void_p(torch.cuda.current_stream().cuda_stream)

  # ctypes argument objects are built ONCE: a call then costs ~2 us of host time, well below the kernels (the first version
  # of this tool rebuilt 22 c_int objects per call and measured the HOST: every column read ~45 us)
  args = (p(dy), p(x), p(ws), p(ss if pro else None), c_int(1), p(slot if pro else None), c_int(8), c_int(M), c_int(N), c_int(C), c_int(k), c_int(k),
          c_int(H), c_int(H), c_int(H), c_int(H), c_int(1), c_int(pad), c_int(pad), c_int(S), c_int64(M), st)
  ts = {}
  for name, lib in libs.items():
    fn = getattr(lib, LAUNCH)
    assert fn(*args) == 0
    ts[name] = timeit(lambda: fn(*(args[:-1] + (c_void_p(torch.cuda.current_stream().cuda_stream),))))
  red = getattr(prod, '_Z13pf_wrw_reducePfilPviP12ihipStream_t')
  rargs = (p(ws), c_int(S), c_int64(N * taps * C), p(dw), c_int(0), st)
  t_red = timeit(lambda: red(*(rargs[:-1] + (c_void_p(torch.cuda.current_stream().cuda_stream),))))
  slab = S * N * taps * C * 4 / 1e6
  inp = (M * C + M * N) * 2 / 1e6
  print('%-16s %-4d | %7.0f %7.0f %7.0f %7.0f %8.0f | %9.0f | %8.0f %6.0f | %5.0f' % (
      '%d,%d,%d,%d,%d' % (H, C, N, k, pro), S, ts['full'], ts['nomma'], ts['nodma'], ts['nost'], ts['mmaonly'], t_red, slab, inp,
      2.0 * M * N * C * taps / (ts['full'] + t_red) * 1e-6))
