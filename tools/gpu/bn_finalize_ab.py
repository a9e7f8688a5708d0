"""pf_bn_finalize of two builds of the library on the same partial statistics: every output compared, over channel counts and
partial-block counts around the kernel's strides.  usage: python tools/gpu/bn_finalize_ab.py <libA.so> <libB.so>"""
import ctypes, sys
import torch

A, B = ctypes.CDLL(sys.argv[1]), ctypes.CDLL(sys.argv[2])
P = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
worst = 0.0
for training in (1, 0):
  for C in (16, 24, 32, 40, 64, 128, 200, 512, 1024):
    for nb in (1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300, 511, 512, 1000):
      g = torch.Generator(device='cuda').manual_seed(C * 1000 + nb)
      rows = nb * 128
      part = torch.empty(nb, 4, C, device='cuda')
      part[:, 0] = torch.randn(nb, C, device='cuda', generator=g) * 10
      part[:, 1] = torch.rand(nb, C, device='cuda', generator=g) * 300 + 100
      part[:, 2] = -torch.rand(nb, C, device='cuda', generator=g) * 5
      part[:, 3] = torch.rand(nb, C, device='cuda', generator=g) * 5
      x = torch.randn(4, C, device='cuda', generator=g)
      gamma = torch.rand(C, device='cuda', generator=g) + 0.5
      beta = torch.randn(C, device='cuda', generator=g)
      outs = []
      for lib in (A, B):
        mm = torch.linspace(-1, 1, C, device='cuda'); mv = torch.linspace(0.5, 2, C, device='cuda')
        ss = torch.zeros(2, C, device='cuda'); mi = torch.zeros(2, C, device='cuda')
        slot = torch.tensor([0xFFFFFFFF, 0xFFFFFFFF], device='cuda', dtype=torch.int64).to(torch.int32) if False else torch.full((2,), -1, device='cuda', dtype=torch.int32)
        r = lib.pf_bn_finalize(P(part), nb, ctypes.c_int64(rows), C, P(x), 0, P(gamma), P(beta), P(mm), P(mv), ctypes.c_float(0.997),
                               ctypes.c_float(1e-3), training, 1, P(ss), P(mi), P(slot), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert r == 0
        torch.cuda.synchronize()
        outs.append((mm, mv, ss, mi, slot))
      for name, a, b in zip(('moving_mean', 'moving_var', 'scale_shift', 'mean_invstd'), outs[0], outs[1]):
        d = float(((a - b).abs() / (b.abs() + 1e-3)).max())
        worst = max(worst, d)
        if d > 1e-4:
          print('training %d C %d n_blocks %d: %s differs by %.3e' % (training, C, nb, name, d))
      if not torch.equal(outs[0][4], outs[1][4]):
        print('training %d C %d n_blocks %d: slot %s vs %s' % (training, C, nb, outs[0][4].tolist(), outs[1][4].tolist()))
print('worst relative difference %.3e%s' % (worst, '  (bit-identical outputs)' if worst == 0.0 else ''))
