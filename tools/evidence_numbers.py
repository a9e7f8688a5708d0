#!/usr/bin/env python
"""The numbers DESIGN.md section 6 / README quote, read back from profiles/<tag>_*:   python tools/evidence_numbers.py [tag]"""
import csv, importlib.util, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
P = lambda name: os.path.join(ROOT, 'profiles', '%s_%s' % (tag, name))


def line(path):
  return json.loads([l for l in open(path) if l.startswith('{')][0])


d = line(P('bench.json'))
r = d['roofline']
h = r.get('hbm_region')                       # round 6 on: MFMA headline, the HBM region as a sub-block (rounds 1-5: the block itself)
sh = h['shared'] if h else r
print('default: %.0f images/s, %.2f ms/step | step MFMA fraction %.3f | HBM region shared %.0f GB/s = %.3f of peak, %d launches, %.1f us avg; unshared %s | traffic %s | cpu_baseline %.2f img/s (%s) | host submit %s'
      % (d['value'], d['ms_per_step'], r['step_mfma_frac'], sh['achieved'], sh['frac'], sh['launches'], sh['avg_launch_ms'] * 1e3,
         ('%.3f' % (h or r)['unshared']['frac']) if (h or r).get('unshared') else None,
         ('%.1f MB' % ((h or r)['traffic'] / 1e6)) if (h or r).get('traffic') else None, d['cpu_baseline']['value'], d['cpu_baseline']['sample'][-45:-20],
         [round(v, 1) for v in d['host_submit_ms_min_median_max']]))
for c in ('c1', 'c2a32', 'c3', 'c4'):
  if os.path.exists(P('bench_%s.json' % c)):
    x = line(P('bench_%s.json' % c))
    print('%s: %.0f images/s, %.2f ms/step, step MFMA fraction %.3f, host submit median %.1f ms'
          % (c, x['value'], x['ms_per_step'], x['roofline']['step_mfma_frac'], x['host_submit_ms_min_median_max'][1]))
rows = list(csv.reader(open(P('step_kernels_b256.csv'))))
for r_ in rows[:12]:
  if r_ and r_[0].startswith('#') and ('steady' in r_[0] or 'idle' in r_[0] or 'category' in r_[0]):
    print(' '.join(r_))
body = [r_ for r_ in rows if r_ and not r_[0].startswith('#') and r_[0] != 'kernel']
tot = lambda pat: sum(float(r_[4]) for r_ in body if re.match(pat, r_[0]))
for name, pat in (('igemm prologue (mode 2)', r'k_igemm<\d+, \d+, \d+, \d+, \d+, 2[,>]'), ('resident kernel, prologue', r'k_conv1x1_stream<\d+, true'),
                  ('igemm plain (mode 0)', r'k_igemm<\d+, \d+, \d+, \d+, \d+, 0[,>]'), ('igemm backward-data (mode 1)', r'k_igemm<\d+, \d+, \d+, \d+, \d+, 1[,>]'), ('k_wrw2', r'k_wrw2'),
                  ('k_wrw_reduce', r'k_wrw_reduce'), ('k_bn_bwd_apply', r'k_bn_bwd_apply'), ('k_bn_apply', r'k_bn_apply'), ('max-pool', r'k_maxpool'),
                  ('stem', r'k_stem'), ('resident kernel, other modes', r'k_conv1x1_stream<\d+, false'), ('k_conv1x1_fwd', r'k_conv1x1_fwd')):
  print('  %-32s %.2f ms/step' % (name, tot(pat)))
pats = [r'k_conv1x1_stream<\d+, true, ', r'k_igemm<\d+, \d+, \d+, \d+, \d+, 2[,>]', r'k_conv1x1_fwd<\d+, true, ']
t = c = 0
for r_ in csv.DictReader(open(P('rocprofv3_stats_b256.csv'))):
  if any(re.search(p, r_['Name']) for p in pats):
    t += int(r_['TotalDurationNs']); c += int(r_['Calls'])
print('rocprofv3 --stats, region kernels: %d launches, %.1f us average' % (c, t / c / 1e3))
spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
for reg in ('conv1x1_fwd', 'conv1x1_wrw', 'bn_bwd_apply'):
  v = b.pmc_traffic_per_launch(reg, tag)
  print('PMC traffic per launch, %s: %s' % (reg, ('%.1f MB' % (v / 1e6)) if v else None))
print(open(P('pytest_gpu.log')).read().strip().splitlines()[-1])
