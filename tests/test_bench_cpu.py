"""bench.py's control flow end to end WITHOUT a GPU: the HIP entry points are replaced by the float32 emulations of
tests/fake_hip.py and the few torch.cuda calls bench.py makes are stubbed, so that every line of `main()` runs -- argument
handling, learner construction, warm-up, the self-diagnosis (here forced into its host-bound branch), the timed region and
the JSON line with every key of the driver's contract.  (What it measures is meaningless; the GPU runs measure.)"""
import json
import sys

import torch

from fake_hip import FakeHipFull


import pytest


@pytest.mark.parametrize('config,extra', [('c2', []), ('c2a32', []), ('c1', []), ('c4', ['--resnet_size', '18']),
                                          ('c3', ['--image_size', '64'])])
def test_bench_main_runs_end_to_end_on_emulated_kernels(monkeypatch, capsys, tmp_path, config, extra):
  import pocketflow_amd.graph as G
  import pocketflow_amd.learners.channel_pruning.learner as CP
  import pocketflow_amd.plan as P
  import pocketflow_amd.losses as L
  import pocketflow_amd.optim as Opt
  import pocketflow_amd.profiling as PR
  import pocketflow_amd.learners.abstract_learner as AL
  import pocketflow_amd.learners.nonuniform_quantization.utils as NU
  import pocketflow_amd.learners.weight_sparsification.learner as WS
  import pocketflow_amd.learners.layerwise as LW
  import bench
  fake = FakeHipFull()
  for mod in (G, P, L, Opt, WS, NU, LW, CP):
    monkeypatch.setattr(mod, 'hip', fake)
  monkeypatch.setattr(AL, 'require_gpu', lambda: torch.device('cpu'))
  monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
  # pretend the process is host-bound: the diagnostic branch (one more step under cProfile) must work when it is needed
  monkeypatch.setattr(bench, 'launch_probe', lambda torch, n=1000: {'host_us_per_launch': 1.0, 'us_per_dispatch': 99.0})
  monkeypatch.setattr(bench, 'memory_snapshot', lambda torch: {'reserved_gb': 0.0, 'segments_allocated': 0})
  monkeypatch.setattr(PR, 'enable', lambda name: None)
  monkeypatch.setattr(PR, 'summary', lambda name, lo=0, hi=None, side=None: (72, 7.2, 72 * 3.0e8))
  monkeypatch.setenv('TMPDIR', str(tmp_path))
  monkeypatch.setattr(sys, 'argv', ['bench.py', '--config', config, '--steps', '1', '--warmup', '1', '--batch', '2',
                                    '--image_size', '32', '--dtype', 'float32', '--no_cpu_baseline'] + extra)
  from pocketflow_amd.flags import FLAGS
  try:
    bench.main()
  finally:
    FLAGS.reset()
  out, err = capsys.readouterr()
  line = json.loads([ln for ln in out.splitlines() if ln.startswith('{')][-1])
  for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
    assert key in line, key
  assert line['n_gpus'] == 1 and line['steps'] == 1 and line['warmup'] == 1 and line['scaling'] == 'weak'
  assert line['higher_is_better'] is True and line['vs_baseline'] is None and line['data'] == 'synthetic'
  assert line['unit'] == 'images/s' and line['value'] > 0 and 'workload' in line['config'] and 'model' not in line['config']
  assert line['config']['name'] == config
  want = {'c2': 'UniformQuantLearner w8/a8', 'c2a32': 'w8/a32', 'c1': 'WeightSparseLearner', 'c3': 'ChannelPrunedLearner',
          'c4': 'NonUniformQuantLearner 4-bit'}[config]
  assert want in line['config']['workload'], line['config']['workload']
  r = line['roofline']
  # headline: the MFMA roofline north_star names (whole-step FLOPs / step time / dense bf16 peak); the HBM-bound region measured launch
  # by launch is a sub-block, the figure with the chip to itself first
  assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and r['traffic'] is None
  assert abs(r['frac'] - r['step_mfma_frac']) < 1e-12 and abs(r['achieved'] * 1e12 - line['value'] * bench.CONFIGS[config]['flops']) < 1e-3 * r['achieved'] * 1e12
  h = r['hbm_region']
  assert h['bound'] == 'hbm' and h['unit'] == 'GB/s' and list(h)[-2:] == ['unshared', 'shared'] and h['shared']['launches'] == 72
  assert abs(h['shared']['frac'] - h['shared']['achieved'] / h['peak']) < 1e-9
  assert line['launch_probe']['after_warmup']['us_per_dispatch'] == 99.0 and 'memory' in line
  assert 'host-bound process' in err and 'tottime' in err          # the self-diagnosis printed its profile


@pytest.mark.parametrize('config', ['c1', 'c2', 'c2a32', 'c3', 'c4'])
def test_bench_flag_setup_in_a_fresh_interpreter(config, tmp_path):
  """bench.py assigns reference flags that only exist once the module defining them was imported; in a fresh process
  (how the driver runs it) nothing else has imported them.  (Round 3, GPU call 2: `Unknown command line flag
  'enbl_multi_gpu'` -- the in-process test above had passed because other tests' imports had defined the flag.)"""
  import os
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  code = ('import sys; sys.argv = ["bench.py", "--config", %r]; sys.path.insert(0, %r); import bench; '
          'a = bench.parse_args(); mh, ln = bench.set_flags(a, %r, 1); '
          'from pocketflow_amd.flags import FLAGS; print("OK", mh.__name__, ln.__name__, FLAGS.batch_size, FLAGS.enbl_multi_gpu)'
          % (config, root, str(tmp_path)))
  out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
  assert out.returncode == 0 and 'OK ModelHelper' in out.stdout, out.stderr[-2000:]
