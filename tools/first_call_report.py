"""One screen from the outputs of tools/gpu/next_round_first_call.sh (gpurun_out/r04_first_*): the step with every prepared variant
against the product library in the same box, the per-layer A/B with its bit-identity column, the test tails, the ablation and
stagger tables.      python tools/first_call_report.py [gpurun_out] [r04_first]"""
import glob
import json
import os
import sys


def bench_rows(d, tag):
  rows = []
  for path in sorted(glob.glob(os.path.join(d, tag + '_bench_*.json'))):
    name = os.path.basename(path)[len(tag) + 7:-5]
    line = None
    for ln in open(path):
      if ln.startswith('{'):
        line = json.loads(ln)
    if line is None:
      rows.append((name, None))
      continue
    r = line.get('roofline') or {}
    rows.append((name, dict(value=line['value'], ms=line['ms_per_step'], frac=r.get('frac'), region_ms=r.get('avg_launch_ms'),
                            launches=r.get('launches'), host=line.get('host_submit_ms_per_step'),
                            teacher=(line.get('config') or {}).get('teacher'), config=(line.get('config') or {}).get('name'))))
  return rows


def main(argv):
  d = argv[1] if len(argv) > 1 else 'gpurun_out'
  tag = argv[2] if len(argv) > 2 else 'r04_first'
  rows = bench_rows(d, tag)
  base = [r for n, r in rows if r and n.startswith('product') and r.get('config') in (None, 'c2')]
  ref = sum(r['value'] for r in base) / len(base) if base else None
  print('%-14s %10s %9s %8s %10s %9s %8s  %s' % ('bench', 'images/s', 'ms/step', 'vs prod', 'roofline', 'region ms', 'host ms', 'teacher'))
  for n, r in rows:
    if r is None:
      print('%-14s (no JSON line: see the .err file)' % n)
      continue
    rel = ('%.3f' % (r['value'] / ref)) if (ref and r.get('config') in (None, 'c2')) else ''
    print('%-14s %10.0f %9.2f %8s %10s %9s %8s  %s' % (
        n, r['value'], r['ms'], rel, '%.3f' % r['frac'] if r['frac'] is not None else '-',
        '%.4f' % r['region_ms'] if r['region_ms'] else '-', '%.1f' % r['host'] if r['host'] else '-', r['teacher'] or ''))
  for suffix in ('igemm_sgb.txt', 'pytest_sgb.log', 'pytest_sgb2.log', 'pytest_ahead.log', 'igemm_ablation.txt', 'igemm_stagger.txt',
                 'fwd1x1_layers.txt', 'fwd1x1_layers_sgb.txt', 'fwd1x1_layers_sgb2.txt', 'wrw.txt'):
    path = os.path.join(d, '%s_%s' % (tag, suffix))
    print('\n== %s' % os.path.basename(path))
    if not os.path.exists(path):
      print('(missing)')
      continue
    for ln in open(path).read().splitlines()[-60:]:
      print(ln[:200])


if __name__ == '__main__':
  main(sys.argv)
