"""GPU time of a kernel launcher, free of host launch overhead: n calls are captured into ONE hipGraph (after a warm-up call
outside the capture: first calls configure dynamic LDS sizes) and the replay is timed with events.  Falls back to plain
back-to-back launches if the capture fails."""
import torch


def gpu_time_us(fn, n=20, reps=3):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
      for _ in range(n):
        fn()
    run = g.replay
  except Exception as e:  # pylint: disable=broad-except
    print('# graph capture failed (%s): timing plain launches' % e)

    def run():
      for _ in range(n):
        fn()
  run()
  torch.cuda.synchronize()
  best = float('inf')
  for _ in range(reps):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    run()
    b.record()
    torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) / n * 1e3)
  return best
