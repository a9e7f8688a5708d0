"""Where does the time of the implicit-GEMM kernel go?  pf_conv2d_fwd of the product library next to three ABLATION builds of
pf_igemm.hip (tools/gpu/build_ablate.sh -> tools/gpu/_build/libig_ablate{1,2,3}.so):
  full  the product kernel
  fill  -DPF_IG_ABLATE=1: LDS-DMA + barriers only (no fragment reads, no MFMAs)
  comp  -DPF_IG_ABLATE=2: fragment reads + MFMAs + barriers (no LDS-DMA)
  mfma  -DPF_IG_ABLATE=3: MFMAs + barriers (no LDS-DMA, no fragment reads)
  mf-e  -DPF_IG_ABLATE=4: as mfma, and NO EPILOGUE (no C staging, statistics, stores): the matrix work of the main loop alone
  noepi -DPF_IG_ABLATE=5: the complete main loop, no epilogue (full - noepi = what the epilogue costs in the launch)
'ideal' = ceil(tiles / workgroup slots) x workgroups per CU x flops of one tile / the CU's share of the 2.5 PFLOP/s peak: what the
matrix pipe of the busiest CU needs for the tile schedule as it is (tile quantisation included).  Per ResNet-50 shape (B = 256) and tile (PF_IGEMM_TILE).  LDS-fill traffic per launch = tiles x k-steps x stage bytes is printed
beside the times: 'fill TB/s' = that traffic / the fill-only time."""
import ctypes, os, sys
from ctypes import c_int, c_void_p
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip

here = os.path.dirname(os.path.abspath(__file__))
libs = {'full': hip._lib}
for n, name in ((1, 'fill'), (2, 'comp'), (3, 'mfma'), (4, 'mf-e'), (5, 'noepi')):
  libs[name] = ctypes.CDLL(os.path.join(here, '_build', 'libig_ablate%d.so' % n))


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import gpu_time_us as timeit   # hipGraph replay: no host launch overhead in the numbers


def make_args(x, w, y, z, B, H, C, N, k, s, pad, Ho):
  """ctypes argument objects, built ONCE per shape: a call then costs ~2 us of host time (rebuilding them per call makes
  the loop host-bound at 30-45 us and every column reads the same)"""
  st = c_void_p(torch.cuda.current_stream().cuda_stream)
  p = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
  return (p(x), p(w), p(y), p(z), p(None), p(None), p(None), p(None), p(None), c_int(0), c_int(B), c_int(H), c_int(H),
          c_int(C), c_int(N), c_int(k), c_int(k), c_int(s), c_int(pad), c_int(pad), c_int(Ho), c_int(Ho), st)


B = int(os.environ.get('B', 256))
shapes = [(56, 64, 64, 3, 1), (28, 128, 128, 3, 1), (14, 256, 256, 3, 1), (7, 512, 512, 3, 1), (14, 1024, 256, 1, 1), (14, 256, 1024, 1, 1)]
tiles = os.environ.get('TILES', '128x128,256x128,128x64').split(',')
print('%-18s %-8s | %8s %8s %8s %8s %8s %8s %8s | %9s %9s | %s' % ('H,C,N,k,s', 'tile', 'full us', 'fill us', 'comp us', 'mfma us', 'mf-e us', 'noepi us', 'ideal us', 'fill MB', 'fill TB/s', 'TF full'))
for H, C, N, k, s in shapes:
  g = torch.Generator(device='cuda').manual_seed(H + C + N)
  x = torch.randn(B, H, H, C, device='cuda', generator=g).bfloat16()
  w = (torch.randn(N, k, k, C, device='cuda', generator=g) * 0.05).bfloat16()
  pad = (k - 1) // 2
  Ho = (H + 2 * pad - k) // s + 1
  y = torch.empty(B, Ho, Ho, N, device='cuda', dtype=torch.bfloat16)
  z = hip.zero_page(x.device)
  M = B * Ho * Ho
  for t in tiles:
    bm, bn = (int(v) for v in t.split('x'))
    if N % bn:
      continue
    os.environ['PF_IGEMM_TILE'] = t
    hip.tuning_reload()          # the library reads its switches once
    args = make_args(x, w, y, z, B, H, C, N, k, s, pad, Ho)
    ts = {}
    for name, lib in libs.items():
      fn = lib.pf_conv2d_fwd
      assert fn(*args) == 0
      ts[name] = timeit(lambda: fn(*(args[:-1] + (c_void_p(torch.cuda.current_stream().cuda_stream),))))
    ntile = ((M + bm - 1) // bm) * (N // bn)
    steps = k * k * C // 64
    mb = ntile * steps * (bm + bn) * 128 / 1e6
    slots = 256 if bm == 256 else 512                       # ig_pick: resident workgroups on the chip
    ideal = -(-ntile // slots) * (slots // 256) * 2.0 * bm * bn * C * k * k / (2.5e15 / 256) * 1e6   # us: rounds x workgroups per CU x tile flops / CU peak
    print('%-18s %-8s | %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f %8.1f | %9.0f %9.1f | %5.0f' % (
        '%d,%d,%d,%d,%d' % (H, C, N, k, s), t, ts['full'], ts['fill'], ts['comp'], ts['mfma'], ts['mf-e'], ts['noepi'], ideal, mb, mb / ts['fill'],
        2.0 * M * N * C * k * k / ts['full'] * 1e-6))
os.environ.pop('PF_IGEMM_TILE', None)
hip.tuning_reload()          # the library reads its switches once
