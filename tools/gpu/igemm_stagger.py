"""Two workgroups share a CU in the plain implicit-GEMM kernels (128-row tiles, 2 x 64 KiB of LDS); identical tiles keep them in
the SAME phase, so both are in their epilogue -- no matrix work -- at the same time.  PF_IGEMM_STAGGER_MODE=1 starts the second
half of the grid PF_IGEMM_STAGGER clocks late; this table is the 3x3 forward layers of ResNet-50 (B = 256) and the deep plain 1x1
GEMMs of backward-data against that delay (0 = product).  hipGraph-replay timing, us."""
import os, sys
from ctypes import c_int, c_void_p
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import gpu_time_us as timeit

B = int(os.environ.get('B', 256))
delays = [int(v) for v in os.environ.get('DELAYS', '0,1500,3000,4500,6000,9000,12000').split(',')]
shapes = [(56, 64, 64, 3), (28, 128, 128, 3), (14, 256, 256, 3), (7, 512, 512, 3), (28, 512, 128, 1), (14, 1024, 256, 1), (14, 256, 1024, 1), (7, 2048, 512, 1)]
fn = hip._lib.pf_conv2d_fwd
p = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
print('%-16s | %s' % ('H,C,N,k', ' '.join('%7d' % d for d in delays)))
os.environ['PF_IGEMM_STAGGER_MODE'] = '1'
hip.tuning_reload()          # the library reads its switches once
for H, C, N, k in shapes:
  g = torch.Generator(device='cuda').manual_seed(H + C + N)
  x = torch.randn(B, H, H, C, device='cuda', generator=g).bfloat16()
  w = (torch.randn(N, k, k, C, device='cuda', generator=g) * 0.05).bfloat16()
  pad = (k - 1) // 2
  y = torch.empty(B, H, H, N, device='cuda', dtype=torch.bfloat16)
  z = hip.zero_page(x.device)
  ref = None
  row = []
  for d in delays:
    os.environ['PF_IGEMM_STAGGER'] = str(d)
    hip.tuning_reload()          # the library reads its switches once
    def call():
      st = c_void_p(torch.cuda.current_stream().cuda_stream)
      assert fn(p(x), p(w), p(y), p(z), p(None), p(None), p(None), p(None), p(None), c_int(0), c_int(B), c_int(H), c_int(H), c_int(C),
                c_int(N), c_int(k), c_int(k), c_int(1), c_int(pad), c_int(pad), c_int(H), c_int(H), st) == 0
    call()
    torch.cuda.synchronize()
    if ref is None:
      ref = y.clone()
    else:
      assert torch.equal(ref, y), 'a delayed start changed the result'
    row.append(timeit(call))
  print('%-16s | %s' % ('%d,%d,%d,%d' % (H, C, N, k), ' '.join('%7.1f' % t for t in row)))
os.environ.pop('PF_IGEMM_STAGGER', None)
hip.tuning_reload()          # the library reads its switches once
os.environ.pop('PF_IGEMM_STAGGER_MODE', None)
hip.tuning_reload()          # the library reads its switches once
