"""Diagnostics: bench.py with the phases of every recorded-step replay timed on the host (graph launches / exchange call), per rank.
python -m torch.distributed.run ... tools/gpu/replay_timing_wrap.py <bench.py arguments>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from pocketflow_amd import step_graph  # noqa: E402

LOG = []


def replay(self):
  t = [time.perf_counter()]
  for i, g in enumerate(self.graphs):
    g.replay()
    t.append(time.perf_counter())
    if i < len(self.actions):
      self.actions[i]()
      t.append(time.perf_counter())
  torch.cuda.synchronize()
  t.append(time.perf_counter())
  LOG.append([round((b - a) * 1e3, 2) for a, b in zip(t[:-1], t[1:])])
  return None


step_graph.CudaBackend.replay = replay
import bench  # noqa: E402

try:
  bench.main()
finally:
  sys.stderr.write('REPLAY_PHASES rank %s (ms: graph 0 launch, exchange call, graph 1 launch, device drain): %s\n' % (os.environ.get('RANK', '0'), LOG[-8:]))
