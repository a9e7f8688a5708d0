#!/usr/bin/env python
"""Host-side view of the steady state: HIP API calls of a rocprofv3 `--hip-trace --kernel-trace --output-format csv` run inside the
window of the last K steps (same window as tools/prof_summary.py: between fused-Adam launches), per API name: calls per step,
host time per step, longest call -- and what the host was doing during the longest device idle gaps.

    python tools/hip_api_summary.py <..._hip_api_trace.csv> <..._kernel_trace.csv> --steps 4"""
import argparse
import csv
from collections import defaultdict


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('api')
  ap.add_argument('kernels')
  ap.add_argument('--steps', type=int, default=4)
  ap.add_argument('--marker', default='k_adam_flat<unsigned short')
  args = ap.parse_args()
  ks = []
  with open(args.kernels, newline='') as f:
    for r in csv.DictReader(f):
      ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
  ks.sort()
  marks = [e for s, e, n in ks if args.marker in n]
  K = min(args.steps, len(marks) - 1)
  t0, t1 = marks[-K - 1], marks[-1]
  api = []
  with open(args.api, newline='') as f:
    for r in csv.DictReader(f):
      api.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Function') or r.get('Name') or r.get('Operation', '?')))
  api.sort()
  agg = defaultdict(lambda: [0, 0, 0])
  for s, e, n in api:
    if s >= t0 and e <= t1:
      a = agg[n]
      a[0] += 1
      a[1] += e - s
      a[2] = max(a[2], e - s)
  print('# host API calls inside the device window of %d steps (%.2f ms/step)' % (K, (t1 - t0) / K / 1e6))
  print('api,calls_per_step,host_ms_per_step,longest_us')
  for n, (c, ns, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%s,%.1f,%.3f,%.1f' % (n, c / K, ns / K / 1e6, mx / 1e3))
  # device idle gaps and the API calls overlapping them
  gaps, horizon, prev = [], None, None
  for s, e, n in ks:
    if s < t0 or e > t1:
      continue
    if horizon is not None and s > horizon + 20000:
      gaps.append((s - horizon, horizon, s, prev, n))
    if horizon is None or e > horizon:
      horizon, prev = e, n
  for d, g0, g1, a, b in sorted(gaps, key=lambda g: -g[0])[:8]:
    during = [(n, (min(e, g1) - max(s, g0)) / 1e3) for s, e, n in api if e > g0 and s < g1]
    during.sort(key=lambda x: -x[1])
    print('# gap %.0f us between %s -> %s | host during it: %s' % (
        d / 1e3, a[:50], b[:50], ', '.join('%s %.0f us' % x for x in during[:4])))


if __name__ == '__main__':
  main()
