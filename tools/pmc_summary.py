#!/usr/bin/env python
"""Per-kernel mean of one rocprofv3 PMC counter (counter_collection.csv), for the pocketflow_hip kernels.

    python tools/pmc_summary.py <..._counter_collection.csv> FETCH_SIZE --out profiles/r01_pmc_fetch.csv
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE
under-reports wide coalesced reads by 2x on gfx950 -- the correction is applied by the consumer
(bench.py / DESIGN.md), this file keeps the raw values."""
import argparse
import csv
import re
import sys
from collections import defaultdict


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('csv')
  ap.add_argument('counter')
  ap.add_argument('--prefix', default='k_')
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  agg = defaultdict(lambda: [0, 0.0, 0.0])
  with open(args.csv, newline='') as f:
    for r in csv.DictReader(f):
      if r.get('Counter_Name') != args.counter:
        continue
      name = re.sub(r'^void ', '', r['Kernel_Name'])
      if not name.startswith(args.prefix):
        continue
      name = re.sub(r'\(.*$', '', name)
      a = agg[name]
      v = float(r['Counter_Value'])
      a[0] += 1
      a[1] += v
      a[2] = max(a[2], v)
  out = open(args.out, 'w', newline='') if args.out else sys.stdout
  w = csv.writer(out)
  w.writerow(['kernel', 'dispatches', 'mean_' + args.counter, 'max_' + args.counter])
  for k, (n, s, m) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([k, n, '%.3f' % (s / n), '%.3f' % m])
  if args.out:
    out.close()
    print(open(args.out).read())


if __name__ == '__main__':
  main()
