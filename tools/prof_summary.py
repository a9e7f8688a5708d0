#!/usr/bin/env python
"""Reduce a rocprofv3 `--kernel-trace --output-format csv` trace to a steady-state per-kernel summary.

A whole-run `--stats` table is dominated by MIOpen's one-off solver search during warm-up; what the
roofline figures need is the steady state.  The fine-tune step launches the fused Adam kernel over the
matmul-kernel buffer exactly once per step, so the window between the (K+1)-th last and the last such
launch holds exactly K steps.

    python tools/prof_summary.py <..._kernel_trace.csv> --steps 4 --out profiles/r01_step_kernels.csv
"""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
  name = re.sub(r'\(.*$', '', name)                # drop the argument list
  name = re.sub(r'^void ', '', name)
  return name[:150]


def category(name: str) -> str:
  if name.startswith('k_') or name.startswith('void k_'):
    return 'pocketflow_hip'
  low = name.lower()
  if 'igemm' in low or 'ck::' in low or '_zn2ck' in low or 'cijk' in low or 'conv' in low or 'gemm' in low \
      or 'subtensorop' in low or 'batched_transpose' in low:
    return 'miopen/blas conv+gemm'
  if 'nccl' in low or 'rccl' in low:
    return 'rccl'
  return 'torch aten'


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('trace')
  ap.add_argument('--steps', type=int, default=4)
  ap.add_argument('--marker', default='k_adam_flat<unsigned short')
  ap.add_argument('--marker_fallback', default='k_adam_flat')
  ap.add_argument('--out', default=None)
  ap.add_argument('--top', type=int, default=60)
  args = ap.parse_args()
  rows = []
  with open(args.trace, newline='') as f:
    rd = csv.DictReader(f)
    for r in rd:
      rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
  rows.sort()
  marks = [e for s, e, n in rows if args.marker in n]
  if len(marks) < 2:
    marks = [e for s, e, n in rows if args.marker_fallback in n]
    # two launches per step in that case (W and O buffers)
    marks = marks[1::2]
  # several marker launches per step (the optimiser updates the kernel buffer and the small buffer back to back; C3's table of
  # round 4 was per HALF step because of it): keep the last mark of every cluster -- marks closer than a quarter of a typical step
  # (the median of the larger half of the distances between neighbours) belong to one step
  if len(marks) > 2:
    dist = sorted(b - a for a, b in zip(marks[:-1], marks[1:]))
    upper = dist[len(dist) // 2:]
    near = upper[len(upper) // 2] / 4.0
    marks = [m for i, m in enumerate(marks) if i == len(marks) - 1 or marks[i + 1] - m > near]
  K = min(args.steps, len(marks) - 1)
  if K < 1:
    sys.exit('marker kernel %r not found often enough' % args.marker)
  # the K CONSECUTIVE steps with the smallest wall time: bench.py runs a 1000-launch empty-kernel probe (`launch_probe`: aten add_ on
  # 64 floats) and mode hand-overs between some of its steps -- the LAST K steps of the trace (rounds 1-5) contained that probe, which
  # showed as "250 aten add_ launches per step" in the step tables (they were never part of a step)
  best = min(range(K, len(marks)), key=lambda i: marks[i] - marks[i - K])
  t0, t1 = marks[best - K], marks[best]
  agg = defaultdict(lambda: [0, 0])
  busy = 0
  for s, e, n in rows:
    if s >= t0 and e <= t1:
      a = agg[n]
      a[0] += 1
      a[1] += e - s
      busy += e - s
  wall = (t1 - t0) / K
  # idle time of the device inside the window: gaps between the end of everything launched so far and the next start
  gaps, horizon, prev = [], None, None
  for s, e, n in rows:
    if s < t0 or e > t1:
      continue
    if horizon is not None and s > horizon:
      gaps.append((s - horizon, prev, n))
    if horizon is None or e > horizon:
      horizon, prev = e, n
  idle = sum(g[0] for g in gaps)
  lines = []
  cats = defaultdict(int)
  for n, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    cats[category(n)] += ns
    lines.append((short(n), category(n), c / K, ns / c / 1e3, ns / K / 1e6, 100.0 * ns / busy))
  out = open(args.out, 'w', newline='') if args.out else sys.stdout
  w = csv.writer(out)
  w.writerow(['# steady state over %d steps: wall %.3f ms/step, GPU busy %.3f ms/step' % (K, wall / 1e6, busy / K / 1e6)])
  w.writerow(['# idle between kernels: %.3f ms/step in %.0f gaps/step (median %.1f us); gaps > 20 us: %d/step, %.3f ms/step'
              % (idle / K / 1e6, len(gaps) / K, (sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3) if gaps else 0.0,
                 sum(1 for g in gaps if g[0] > 20000) / K, sum(g[0] for g in gaps if g[0] > 20000) / K / 1e6)])
  for d, a, b in sorted(gaps, key=lambda g: -g[0])[:6]:
    w.writerow(['# gap', '%.1f us' % (d / 1e3), short(a)[:60], '->', short(b)[:60]])
  for c, ns in sorted(cats.items(), key=lambda kv: -kv[1]):
    w.writerow(['# category', c, '%.3f ms/step' % (ns / K / 1e6), '%.1f %%' % (100.0 * ns / busy)])
  w.writerow(['kernel', 'category', 'calls_per_step', 'avg_us', 'ms_per_step', 'pct_of_busy'])
  for l in lines[:args.top]:
    w.writerow([l[0], l[1], '%.2f' % l[2], '%.2f' % l[3], '%.4f' % l[4], '%.2f' % l[5]])
  rest = lines[args.top:]
  if rest:
    w.writerow(['(%d more kernels)' % len(rest), '', '%.2f' % sum(l[2] for l in rest), '',
                '%.4f' % sum(l[4] for l in rest), '%.2f' % sum(l[5] for l in rest)])
  if args.out:
    out.close()
    print(open(args.out).read()[:6000])


if __name__ == '__main__':
  main()
