#!/usr/bin/env python
"""Matrix-pipe utilisation per kernel from one rocprofv3 PMC pass (north_star: "rocprof MFMA-busy").

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python bench.py --steps 2 ...
    python tools/pmc_table.py <..._counter_collection.csv> > profiles/rNN_pmc_mfma_busy.csv
    python tools/mfma_busy.py profiles/rNN_pmc_mfma_busy.csv [profiles/rNN_step_kernels_b256.csv]

What the counters are (MI355X_MICROARCH.md, PMC section; ROCm 7.2 has no gfx950 derived-counter definitions, so nothing is taken from
`MfmaUtil`):
  SQ_VALU_MFMA_BUSY_CYCLES  cycles in which a SIMD's matrix pipe is busy, summed over every SIMD that executed the dispatch
                            (the guide: = 32 x N_mfma for v_mfma_f32_32x32x16_bf16, i.e. the instruction's issue cycles on its SIMD)
  GRBM_GUI_ACTIVE           cycles the graphics engine was active during the dispatch (elapsed shader-clock cycles; rocprofv3 reports
                            the sum over the 8 XCDs)
  utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
Calibration printed with the table: for the plain implicit-GEMM kernels the number of MFMA instructions per dispatch is known from
the shapes (flops / 16 384 for v_mfma_f32_16x16x32_bf16), so BUSY / N_mfma must come out near the instruction's 16 issue cycles if
the reading above is right; with the step table given, the utilisation is also computed against the traced kernel time at the
nominal 2.4 GHz (an upper bound on the clock: the chip runs 1.9-2.3 GHz under matrix load)."""
import csv
import sys


def main():
  rows = list(csv.DictReader(open(sys.argv[1])))
  dur = {}
  if len(sys.argv) > 2:
    for r in csv.reader(open(sys.argv[2])):
      if len(r) >= 5 and r[0] and not r[0].startswith('#') and r[0] != 'kernel':
        try:
          dur[r[0]] = (float(r[2]), float(r[3]))          # calls per step, average us
        except ValueError:
          pass
  out = []
  for r in rows:
    k = r['kernel']
    try:
      busy = float(r.get('mean_SQ_VALU_MFMA_BUSY_CYCLES', r.get('SQ_VALU_MFMA_BUSY_CYCLES')))      # (pmc_table.py: per-dispatch means)
      gui = float(r.get('mean_GRBM_GUI_ACTIVE', r.get('GRBM_GUI_ACTIVE')))
    except (KeyError, ValueError, TypeError):
      continue
    if busy <= 0:
      continue
    util = busy / (1024.0 * gui / 8.0) if gui > 0 else float('nan')
    util_t = float('nan')
    if k in dur:
      util_t = busy / (1024.0 * dur[k][1] * 1e-6 * 2.4e9)
    out.append((busy * float(r.get('dispatches', 1) or 1), k, int(float(r.get('dispatches', 0) or 0)), busy, gui, util, util_t))
  out.sort(reverse=True)
  print('%-64s %6s %14s %14s %10s %12s' % ('kernel', 'disp', 'MFMA busy cyc', 'GUI active', 'util', 'util @2.4GHz'))
  for _, k, d, busy, gui, util, util_t in out:
    print('%-64s %6d %14.0f %14.0f %9.1f%% %11.1f%%' % (k[:64], d, busy, gui, 100 * util, 100 * util_t))
  if dur:
    # The WHOLE step: every kernel's per-dispatch busy cycles x its calls per step (step table), against 1024 SIMDs x the step's
    # time.  Two denominators: the step's wall time (what "fraction of the MFMA roofline" means for the job) and the GPU-busy time
    # (kernel time summed over both streams); the clock is the nominal 2.4 GHz, an upper bound (1.9-2.3 GHz under matrix load), so
    # both figures are LOWER bounds of the pipe's busy fraction at the real clock.
    wall = gpu_busy = None
    for ln in open(sys.argv[2]):
      if ln.startswith('"# steady state') or ln.startswith('# steady state'):
        import re
        m = re.search(r'wall ([0-9.]+) ms/step, GPU busy ([0-9.]+) ms/step', ln)
        if m:
          wall, gpu_busy = float(m.group(1)), float(m.group(2))
    tot = 0.0
    missing = []
    for _, k, d, busy, gui, util, util_t in out:
      if k in dur:
        tot += busy * dur[k][0]
      else:
        missing.append(k)
    print()
    print('whole step: sum over kernels of (MFMA busy cycles per dispatch x calls per step) = %.4e SIMD-cycles per step' % tot)
    if wall:
      print('  / (1024 SIMDs x %.3f ms wall x 2.4 GHz)     = %.1f %% of the matrix pipes busy over the step' % (wall, 100 * tot / (1024 * wall * 1e-3 * 2.4e9)))
      print('  / (1024 SIMDs x %.3f ms GPU-busy x 2.4 GHz) = %.1f %%' % (gpu_busy, 100 * tot / (1024 * gpu_busy * 1e-3 * 2.4e9)))
      print('  (a 16x16x32 bf16 MFMA holds its pipe 16 cycles for 16 384 flop: the same figure from the flops is')
      print('   flops per step / 16 384 x 16 / (1024 x wall x 2.4e9) = roofline.step_mfma_frac of bench.py, which divides by the 2.5 PFLOP/s peak)')
    if missing:
      print('  kernels with MFMA cycles in the PMC pass but absent from the step table (not counted): %s' % ', '.join(m[:40] for m in missing[:6]))


if __name__ == '__main__':
  main()
