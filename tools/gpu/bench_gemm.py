"""Micro-benchmark: hand-written MFMA GEMM vs torch.mm (hipBLASLt) vs F.conv2d (MIOpen) on the 1x1
convolution shapes of ResNet-50 at batch 256 (NHWC => plain GEMMs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from pocketflow_amd import hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shapes = [(56, 64, 256), (56, 64, 64), (56, 256, 64), (56, 256, 128), (28, 128, 512), (28, 512, 128),
          (28, 512, 256), (14, 256, 1024), (14, 1024, 256), (14, 1024, 512), (7, 512, 2048), (7, 2048, 512)]


def timeit(fn, n=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n):
    fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n


print('%-22s %10s %10s %10s | %10s %10s | %10s %10s   (ms; TF/s of mine)' %
      ('HxW,Cin->Cout', 'mine_nt', 'torch.mm', 'conv2d', 'mine_nn', 'mm_nn', 'mine_tn', 'mm_tn'))
for hw, cin, cout in shapes:
  M = B * hw * hw
  X = torch.randn(M, cin, device='cuda').to(torch.bfloat16)
  W = torch.randn(cout, cin, device='cuda').to(torch.bfloat16)
  Y = torch.empty(M, cout, device='cuda', dtype=torch.bfloat16)
  dY = torch.randn(M, cout, device='cuda').to(torch.bfloat16)
  dX = torch.empty(M, cin, device='cuda', dtype=torch.bfloat16)
  dW = torch.zeros(cout, cin, device='cuda', dtype=torch.float32)
  x4 = X.view(B, hw, hw, cin).permute(0, 3, 1, 2)
  w4 = W.view(cout, 1, 1, cin).permute(0, 3, 1, 2)
  t_mine = timeit(lambda: hip.gemm_bf16_nt(X, W, Y, M, cout, cin))
  t_mm = timeit(lambda: torch.mm(X, W.t(), out=Y))
  t_conv = timeit(lambda: F.conv2d(x4, w4))
  t_nn = timeit(lambda: hip.gemm_bf16_nn(dY, W, dX, M, cin, cout))
  t_mmnn = timeit(lambda: torch.mm(dY, W, out=dX))
  t_tn = timeit(lambda: hip.gemm_bf16_tn(dY, X, dW, cout, cin, M))
  t_mmtn = timeit(lambda: torch.mm(dY.t(), X))
  fl = 2.0 * M * cin * cout
  print('%3dx%-3d %5d->%-5d %10.3f %10.3f %10.3f | %10.3f %10.3f | %10.3f %10.3f   (%.0f / %.0f / %.0f TF/s)' %
        (hw, hw, cin, cout, t_mine, t_mm, t_conv, t_nn, t_mmnn, t_tn, t_mmtn,
         fl / t_mine / 1e9, fl / t_nn / 1e9, fl / t_tn / 1e9))
