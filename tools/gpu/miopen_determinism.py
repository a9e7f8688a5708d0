"""Diagnostic: are aten (MIOpen) float32 convolution gradients reproducible call to call when the allocator hands out
blocks with different previous contents?  MobileNet-v1 x0.5 @64, batch 16 layer shapes, channels_last."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
cfgs = [(3, 16, 3, 2, 1, 64)]
chans = [16, 32, 64, 64, 128, 128, 256, 256, 256, 256, 256, 256, 512, 512]
strides = [1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1]
H = 32
for i, s in enumerate(strides):
  cin, cout = chans[i], chans[i + 1]
  cfgs.append((cin, cin, 3, s, cin, H))          # depthwise
  H = H // s
  cfgs.append((cin, cout, 1, 1, 1, H))           # pointwise


def junk(val):
  t = [torch.full((1 << 14,), val, device='cuda') for _ in range(512)] + [torch.full((1 << 20,), val, device='cuda') for _ in range(16)]
  torch.cuda.synchronize()
  del t


bad = 0
for (cin, cout, k, s, groups, h) in cfgs:
  x = torch.randn(16, cin, h, h, device='cuda').contiguous(memory_format=torch.channels_last)
  w = (torch.randn(cout, cin // groups, k, k, device='cuda') * 0.1).contiguous(memory_format=torch.channels_last)
  pad = (k - 1) // 2
  ho = (h + 2 * pad - k) // s + 1
  dy = torch.randn(16, cout, ho, ho, device='cuda').contiguous(memory_format=torch.channels_last)
  res = []
  for rep, val in enumerate((0.0, 1e3, float('nan'), 0.0)):
    junk(val)
    xx = x.clone().requires_grad_(True); ww = w.clone().requires_grad_(True)
    y = F.conv2d(xx, ww, None, s, pad, 1, groups)
    y.backward(dy)
    res.append((y.detach().clone(), xx.grad.clone(), ww.grad.clone()))
  ref = res[0]
  errs = []
  for r in res[1:]:
    errs.append(tuple(float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)) for a, b in zip(r, ref)))
  flag = any(e > 1e-4 or e != e for t in errs for e in t)
  bad += flag
  print('%s cin %4d cout %4d k %d s %d g %4d H %2d | rel diff vs first call (y, dx, dw): %s' % (
      'BAD' if flag else 'ok ', cin, cout, k, s, groups, h, ' '.join('(%.1e %.1e %.1e)' % t for t in errs)))
print('configurations whose gradients depend on previous memory contents:', bad)
