"""The two reducers behind profiles/ (tools/prof_summary.py, tools/pmc_table.py) on synthetic rocprofv3 CSVs:
the step window is exactly K Adam launches wide, and pmc_table accepts both a raw counter_collection.csv and
one of its own tables (which is what profiles/ holds)."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args):
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', tool)] + list(args), capture_output=True, text=True)
  assert r.returncode == 0, r.stderr
  return r.stdout


def test_prof_summary_takes_exactly_k_steps(tmp_path):
  p = tmp_path / 'x_kernel_trace.csv'
  t = 1000
  with open(p, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Kind', 'Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
    for step in range(6):
      # warm-up junk in the first two steps must stay outside the window
      if step < 2:
        w.writerow(['KERNEL_DISPATCH', 'miopen_search_kernel(int)', t, t + 50000])
        t += 60000
      for _ in range(3):
        w.writerow(['KERNEL_DISPATCH', 'void k_igemm<128, 128, 2, 2, 2, 1>(IgArgs)', t, t + 2000])
        t += 2500
      w.writerow(['KERNEL_DISPATCH', 'void k_adam_flat<unsigned short, true>(AdamArgs)', t, t + 1000])
      t += 1500
  out = _run('prof_summary.py', str(p), '--steps', '3')
  parsed = list(csv.reader(out.strip().splitlines()))
  assert parsed[0][0].startswith('# steady state over 3 steps')
  body = [l for l in parsed if l and not l[0].startswith('#')]
  rows = {r[0]: r for r in body[1:]}
  assert 'miopen_search_kernel' not in rows
  ig = rows['k_igemm<128, 128, 2, 2, 2, 1>']
  assert ig[1] == 'pocketflow_hip' and float(ig[2]) == 3.0 and abs(float(ig[3]) - 2.0) < 1e-6
  assert float(rows['k_adam_flat<unsigned short, true>'][2]) == 1.0


def test_prof_summary_keeps_the_launch_probe_of_bench_py_out_of_the_window(tmp_path):
  """bench.py runs 1000 empty launches (aten add_ on 64 floats) between two of its last steps: the window is the K consecutive
  steps with the smallest wall time, so the probe's step stays outside (rounds 1-5 reported it as 250 aten launches per step)."""
  p = tmp_path / 'x_kernel_trace.csv'
  t = 1000
  with open(p, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Kind', 'Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
    for step in range(9):
      if step == 7:
        for _ in range(100):
          w.writerow(['KERNEL_DISPATCH', 'void at::native::vectorized_elementwise_kernel<4, at::native::CUDAFunctorOnSelf_add<float>, std::array<char*, 2ul> >', t, t + 100])
          t += 300
      for _ in range(3):
        w.writerow(['KERNEL_DISPATCH', 'void k_igemm<128, 128, 2, 2, 2, 1>(IgArgs)', t, t + 2000])
        t += 2500
      w.writerow(['KERNEL_DISPATCH', 'void k_adam_flat<unsigned short, true>(AdamArgs)', t, t + 1000])
      t += 1500
  out = _run('prof_summary.py', str(p), '--steps', '4')
  assert 'CUDAFunctorOnSelf_add' not in out
  parsed = list(csv.reader(out.strip().splitlines()))
  assert parsed[0][0].startswith('# steady state over 4 steps: wall 0.009 ms/step')


def test_prof_summary_groups_two_optimiser_launches_per_step(tmp_path):
  """Adam updates the kernel buffer and the small buffer back to back: with `--marker k_adam_flat` both launches match, and the
  window must still be K STEPS wide (round 4's C3 table was per half step: 9.5 instead of 18.9 ms)."""
  p = tmp_path / 'c3_kernel_trace.csv'
  t = 1000
  with open(p, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Kind', 'Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
    for step in range(7):
      for _ in range(5):
        w.writerow(['KERNEL_DISPATCH', 'void k_dw_fwd<unsigned short, 1>(DwArgs)', t, t + 4000])
        t += 5000
      for buf in ('unsigned short', 'float'):
        w.writerow(['KERNEL_DISPATCH', 'void k_adam_flat<%s, true>(AdamArgs)' % buf, t, t + 1000])
        t += 1200
  out = _run('prof_summary.py', str(p), '--steps', '4', '--marker', 'k_adam_flat')
  parsed = list(csv.reader(out.strip().splitlines()))
  assert parsed[0][0].startswith('# steady state over 4 steps: wall 0.027 ms/step')       # 5 x 5 us + 2 x 1.2 us per step
  row = [r for r in parsed if r and r[0].startswith('k_dw_fwd')][0]
  assert float(row[2]) == 5.0                                                               # calls per STEP, not per half step
  adam = [r for r in parsed if r and r[0].startswith('k_adam_flat')]
  assert sum(float(r[2]) for r in adam) == 2.0


def test_mfma_busy_whole_step_derivation(tmp_path):
  """tools/mfma_busy.py: per-kernel utilisation from one PMC table and, with a step table, the whole-step figure =
  sum(busy cycles per dispatch x calls per step) / (1024 SIMDs x step time x 2.4 GHz) (the derivation must be
  reproducible from the committed CSVs)."""
  pmc = tmp_path / 'pmc.csv'
  with open(pmc, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'dispatches', 'GRBM_GUI_ACTIVE', 'SQ_BUSY_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES'])
    w.writerow(['k_a', 10, 8.0e5, 1e6, 2.4576e7, 1e8])          # 2.4576e7 / (1024 x 8e5 / 8) = 24 %
    w.writerow(['k_b', 4, 8.0e5, 1e6, 0, 1e8])                  # no matrix work: not listed
  steps = tmp_path / 'steps.csv'
  with open(steps, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['# steady state over 4 steps: wall 10.000 ms/step, GPU busy 8.000 ms/step'])
    w.writerow(['kernel', 'category', 'calls_per_step', 'avg_us', 'ms_per_step', 'pct_of_busy'])
    w.writerow(['k_a', 'pocketflow_hip', '100.00', '50.00', '5.0000', '62.5'])
  out = _run('mfma_busy.py', str(pmc), str(steps))
  assert 'k_a' in out and 'k_b' not in out.split('whole step')[0]
  assert '24.0%' in out
  # 2.4576e7 x 100 = 2.4576e9 SIMD-cycles per step; / (1024 x 10 ms x 2.4e9) = 10.0 %, / (1024 x 8 ms x 2.4e9) = 12.5 %
  assert '2.4576e+09 SIMD-cycles per step' in out and '= 10.0 % of the matrix pipes' in out and '= 12.5 %' in out


def test_pmc_table_reads_raw_and_its_own_output(tmp_path):
  raw = tmp_path / 'x_counter_collection.csv'
  with open(raw, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Dispatch_Id', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
    for i, v in enumerate((100.0, 300.0)):
      w.writerow([i, 'void k_wrw2<256, 128, 64, 64, true, 0>(Wrw2Args)', 'FETCH_SIZE', v])
      w.writerow([i, 'void k_wrw2<256, 128, 64, 64, true, 0>(Wrw2Args)', 'WRITE_SIZE', v / 2])
    w.writerow([9, 'at::native::vectorized_elementwise_kernel<4>(int)', 'FETCH_SIZE', 7.0])
  out = _run('pmc_table.py', str(raw))
  rows = list(csv.reader(out.strip().splitlines()))
  assert rows[0] == ['kernel', 'dispatches', 'FETCH_SIZE', 'WRITE_SIZE']
  assert rows[1] == ['k_wrw2<256, 128, 64, 64, true, 0>', '2', '200', '100'] and len(rows) == 2
  table = tmp_path / 'table.csv'
  table.write_text(out)
  again = _run('pmc_table.py', str(table))
  assert list(csv.reader(again.strip().splitlines())) == rows
  bad = tmp_path / 'bad.csv'
  bad.write_text('a,b\n1,2\n')
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_table.py'), str(bad)], capture_output=True,
                     text=True)
  assert r.returncode != 0 and 'Kernel_Name' in r.stderr


def test_profiling_regions_record_event_pairs_and_cost_nothing_when_disabled(monkeypatch):
  """pocketflow_amd/profiling.py with a stand-in for torch.cuda.Event: an enabled region records (start, stop, work) around its
  body -- not when the body raises --, `summary` adds up, and a region that is not enabled is the shared no-op object."""
  import torch
  import pocketflow_amd.profiling as PR

  clock = [0.0]

  class FakeEvent(object):
    def __init__(self, enable_timing=False):
      assert enable_timing
      self.t = None

    def record(self):
      clock[0] += 1.5
      self.t = clock[0]

    def elapsed_time(self, other):
      return other.t - self.t

  monkeypatch.setattr(torch.cuda, 'Event', FakeEvent)
  PR.disable_all()
  try:
    assert PR.region('conv1x1_fwd', 10.0) is PR.region('anything')          # disabled: one shared object, no allocation
    with PR.region('conv1x1_fwd', 10.0):
      pass
    assert PR.summary('conv1x1_fwd') == (0, 0, 0)
    PR.enable('conv1x1_fwd')
    for w in (10.0, 20.0):
      with PR.region('conv1x1_fwd', w):
        clock[0] += 100.0                                                    # the "kernel"
    with PR.region('bn_stats'):                                              # another name stays disabled
      pass
    with PR.suspended():                                                     # launches issued beside other work: not recorded
      assert PR.region('conv1x1_fwd', 7.0) is PR.region('anything')
      with PR.suspended():
        pass
      assert PR.region('conv1x1_fwd', 7.0) is PR.region('anything')
    assert PR.region('conv1x1_fwd', 7.0) is not PR.region('anything')
    PR.include_side = True                                                   # bench.py: the side stream's launches are recorded too, tagged
    with PR.suspended():
      with PR.region('conv1x1_fwd', 1.0):
        clock[0] += 50.0
    PR.include_side = False
    assert PR.summary('conv1x1_fwd', side=True)[0] == 1 and PR.summary('conv1x1_fwd', side=False)[0] == 2
    _ENABLED = PR._enabled['conv1x1_fwd']
    del _ENABLED[-1]                                                         # (the totals below are those of the main-stream launches)
    try:
      with PR.region('conv1x1_fwd', 5.0):
        raise ValueError('launch failed')
    except ValueError:
      pass
    n, ms, work = PR.summary('conv1x1_fwd')
    assert n == 2 and abs(ms - 2 * 101.5) < 1e-9 and work == 30.0
    PR.reset()
    assert PR.summary('conv1x1_fwd') == (0, 0, 0)
  finally:
    PR.disable_all()


def test_isa_diff_tells_identical_changed_removed(tmp_path):
  """tools/isa_diff.py on two hand-made snapshots: label numbering and the per-compilation __hip_cuid symbol do not count, an
  instruction or a descriptor line does."""
  def asm(kernels, cuid):
    out = []
    for idx, (name, body, vgprs) in enumerate(kernels):
      out += ['\t.text', '\t.globl\t%s' % name, '%s:                                 ; @%s' % (name, name)]
      out += ['\ts_load_dword s0, s[4:5], 0x0', '.LBB%d_1:                                ; %%loop' % idx]
      out += ['\t' + i for i in body] + ['\ts_cbranch_scc1 .LBB%d_1' % idx, '\ts_endpgm', '.Lfunc_end%d:' % idx]
      out += ['\t.amdhsa_kernel %s' % name, '\t\t.amdhsa_next_free_vgpr %d' % vgprs, '\t.end_amdhsa_kernel']
    out += ['__hip_cuid_%s:' % cuid, '\t.byte 0', '\t.amdgpu_metadata', 'amdhsa.kernels:', '  - .name: x%s' % cuid, '\t.end_amdgpu_metadata']
    return '\n'.join(out) + '\n'
  a, b = tmp_path / 'a', tmp_path / 'b'
  a.mkdir(); b.mkdir()
  (a / 'k.s').write_text(asm([('k_same', ['v_add_f32 v0, v1, v2'], 8), ('k_gone', ['v_mov_b32 v0, 0'], 4),
                              ('k_code', ['v_add_f32 v0, v1, v2'], 8), ('k_regs', ['v_mul_f32 v0, v1, v2'], 8)], 'aaaa'))
  # k_gone deleted: the remaining kernels move up one label index
  (b / 'k.s').write_text(asm([('k_same', ['v_add_f32 v0, v1, v2'], 8), ('k_code', ['v_fma_f32 v0, v1, v2, v0'], 8),
                              ('k_regs', ['v_mul_f32 v0, v1, v2'], 12)], 'bbbb'))
  (a / 'only_a.s').write_text(asm([('k_x', ['s_nop 0'], 1)], 'cccc'))
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_diff.py'), str(a), str(b)], capture_output=True, text=True)
  assert r.returncode == 1, r.stdout + r.stderr
  assert 'identical   1' in r.stdout and 'changed 2' in r.stdout and 'removed 1' in r.stdout and 'added 0' in r.stdout
  assert 'CHANGED k_code' in r.stdout and 'descriptor same' in r.stdout
  assert 'CHANGED k_regs' in r.stdout and 'descriptor differs' in r.stdout
  assert 'removed k_gone' in r.stdout and 'only_a.s' in r.stdout and 'hip_cuid' not in r.stdout
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_diff.py'), str(a), str(a)], capture_output=True, text=True)
  assert r.returncode == 0 and 'changed 0' in r.stdout
