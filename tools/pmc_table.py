#!/usr/bin/env python
"""All counters of one rocprofv3 --pmc pass as one row per kernel (mean over dispatches):
    python tools/pmc_table.py <..._counter_collection.csv> [--prefix k_]"""
import argparse
import csv
import re
import sys
from collections import defaultdict


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('csv')
  ap.add_argument('--prefix', default='k_')
  args = ap.parse_args()
  agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
  counters = []
  with open(args.csv, newline='') as f:
    rd = csv.DictReader(f)
    if 'Kernel_Name' not in (rd.fieldnames or []):
      if 'kernel' in (rd.fieldnames or []):
        # already one of this tool's tables (what profiles/ holds): echo it, filtered by the prefix
        w = csv.writer(sys.stdout)
        w.writerow(rd.fieldnames)
        for r in rd:
          if r['kernel'].startswith(args.prefix):
            w.writerow([r[k] for k in rd.fieldnames])
        return
      sys.exit('%s: neither a rocprofv3 counter_collection.csv (no Kernel_Name column) nor a pmc_table output'
               % args.csv)
    for r in rd:
      name = re.sub(r'^void ', '', r['Kernel_Name'])
      if not name.startswith(args.prefix):
        continue
      name = re.sub(r'\(.*$', '', name)
      c = r['Counter_Name']
      if c not in counters:
        counters.append(c)
      a = agg[name][c]
      a[0] += 1
      a[1] += float(r['Counter_Value'])
  w = csv.writer(sys.stdout)
  w.writerow(['kernel', 'dispatches'] + counters)
  for k, d in agg.items():
    n = max(v[0] for v in d.values())
    w.writerow([k, n] + ['%.4g' % (d[c][1] / max(d[c][0], 1)) for c in counters])


if __name__ == '__main__':
  main()
