"""Can a replayed hipGraph carry timing events?  torch.cuda.Event(enable_timing=True, external=True) recorded inside a capture becomes an
event-record node; after a replay elapsed_time between two of them should be the enclosed kernels' duration."""
import torch
x = torch.randn(1 << 26, device='cuda')
y = torch.empty_like(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
evs = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(4)]
side = torch.cuda.Stream()
try:
  with torch.cuda.graph(g):
    evs[0].record()
    y.copy_(x); y.mul_(2.0)
    evs[1].record()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      evs[2].record()
      z = x * 3.0
      evs[3].record()
    torch.cuda.current_stream().wait_stream(side)
  for it in range(3):
    g.replay()
    torch.cuda.synchronize()
    print('replay', it, 'main %.3f ms' % evs[0].elapsed_time(evs[1]), 'side %.3f ms' % evs[2].elapsed_time(evs[3]))
  print('GRAPH_EVENTS_OK')
except Exception as e:
  print('GRAPH_EVENTS_FAILED', type(e).__name__, e)
