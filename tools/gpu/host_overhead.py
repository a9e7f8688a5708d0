"""Where the HOST time of a step goes (the CIFAR-size configuration C1 and MobileNet C3 are host-bound): cost of the per-launch helpers
of pocketflow_amd/hip.py and a cProfile of bench.py's C1 step, top functions by own time."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pocketflow_amd import hip


def per_call(fn, n=200000):
  t = time.perf_counter()
  for _ in range(n):
    fn()
  return (time.perf_counter() - t) / n * 1e6


dev = torch.cuda.current_device()
print('hip._stream()                                  %.2f us' % per_call(hip._stream))
print('torch._C._cuda_getCurrentRawStream(dev)        %.2f us' % per_call(lambda: torch._C._cuda_getCurrentRawStream(dev)))
x = torch.zeros(1024, device='cuda')
print('hip._ptr(x)                                    %.2f us' % per_call(lambda: hip._ptr(x)))
print('torch.empty((256, 64), device=cuda, bf16)      %.2f us' % per_call(lambda: torch.empty((256, 64), device='cuda', dtype=torch.bfloat16), 50000))
print('torch.empty_like(x)                            %.2f us' % per_call(lambda: torch.empty_like(x), 50000))
slot = torch.empty(2, dtype=torch.int32, device='cuda')
print('hip.minmax_slots_init(slot) (one tiny launch)  %.2f us' % per_call(lambda: hip.minmax_slots_init(slot), 20000))
torch.cuda.synchronize()

# cProfile of C1 steps through bench.py's own set-up
import importlib.util
spec = importlib.util.spec_from_file_location('bench', os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'bench.py'))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import tempfile
cfgname = os.environ.get('CFG', 'c1')
sys.argv = ['bench.py', '--config', cfgname, '--no_cpu_baseline']
args = b.parse_args()
from pocketflow_amd.flags import FLAGS
tmp = tempfile.mkdtemp(prefix='pf_host_')
learner, step = b.build_learner(args, FLAGS, tmp, 0, 1, lambda: None)
for _ in range(5):
  step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
  step()
host = (time.perf_counter() - t0) / 10 * 1e3
torch.cuda.synchronize()
print('%s: host submits a step in %.2f ms' % (cfgname, host))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
  step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30)
print('\n'.join(l[:160] for l in s.getvalue().splitlines()[:48]))
