"""Micro-benchmark: pf_gemm_bf16_{nt,nn,tn} vs torch (hipBLASLt/rocBLAS) on the ResNet-50 1x1 shapes, B=256."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip

def timeit(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n * 1e-3

B = 256
shapes = [(B*56*56, 64, 64), (B*56*56, 64, 256), (B*56*56, 256, 64), (B*28*28, 256, 128), (B*28*28, 128, 512),
          (B*28*28, 512, 128), (B*14*14, 256, 1024), (B*14*14, 1024, 256), (B*7*7, 512, 2048), (B*7*7, 2048, 512)]
print('%-26s %10s %10s %10s | %10s %10s %10s | hbm-bound us' % ('M,K,N', 'pf nt', 'pf nn', 'pf tn', 'th nt', 'th nn', 'th tn'))
for M, K, N in shapes:
  X = torch.randn(M, K, device='cuda').bfloat16(); W = torch.randn(N, K, device='cuda').bfloat16()
  dY = torch.randn(M, N, device='cuda').bfloat16()
  Y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); dX = torch.empty(M, K, device='cuda', dtype=torch.bfloat16)
  dW = torch.zeros(N, K, device='cuda', dtype=torch.float32)
  t = []
  t.append(timeit(lambda: hip.gemm_bf16_nt(X, W, Y, M, N, K)))
  t.append(timeit(lambda: hip.gemm_bf16_nn(dY, W, dX, M, K, N)))
  t.append(timeit(lambda: hip.gemm_bf16_tn(dY, X, dW, N, K, M)))
  t.append(timeit(lambda: torch.matmul(X, W.t(), out=Y)))
  t.append(timeit(lambda: torch.matmul(dY, W, out=dX)))
  t.append(timeit(lambda: torch.matmul(dY.t(), X)))
  fl = 2.0 * M * K * N
  hbm = (M * K + M * N) * 2 / 8e12 * 1e6
  print('%-26s ' % ('%d,%d,%d' % (M, K, N)) + ' '.join('%7.0fus/%4.0fT' % (x * 1e6, fl / x / 1e12) for x in t) + ' | %6.0f' % hbm)
