#!/usr/bin/env python
"""Fill the @@...@@ placeholders of README.md / profiles/README.md and the results block of DESIGN.md section 6.1 from the evidence
files of a round (profiles/<tag>_*): the documents quote what the files hold, nothing is typed by hand.   python tools/fill_round_results.py r06"""
import csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
P = lambda n: os.path.join(ROOT, 'profiles', '%s_%s' % (tag, n))


def line(path):
  return json.loads([l for l in open(path) if l.startswith('{')][0])


d = line(P('bench.json'))
r = d['roofline']; h = r['hbm_region']
others = {c: line(P('bench_%s.json' % c)) for c in ('c1', 'c2a32', 'c3', 'c4') if os.path.exists(P('bench_%s.json' % c))}
log = open(P('pytest_gpu.log')).read()
m = re.search(r'(\d+) passed(?:, (\d+) skipped)?', log)
passed, skipped = m.group(1), m.group(2) or '0'
rows = list(csv.reader(open(P('step_kernels_b256.csv'))))
head = [' '.join(x) for x in rows[:2]]
body = [x for x in rows if x and not x[0].startswith('#') and x[0] != 'kernel' and len(x) > 4]
tot = lambda pat: sum(float(x[4]) for x in body if re.match(pat, x[0]))
fam = [('fused 1x1 with prologue, three-stage staged kernel `k_igemm<..,3,2>`', r'k_igemm<\d+, \d+, \d+, \d+, 3, 2'),
       ('fused 1x1 with prologue, resident-kernel variant `k_conv1x1_stream<*,true>`', r'k_conv1x1_stream<\d+, true'),
       ('`k_igemm` plain (3x3 forward, prologue-free 1x1)', r'k_igemm<\d+, \d+, \d+, \d+, \d+, 0, false'),
       ('`k_igemm` backward-data with BN sums', r'k_igemm<\d+, \d+, \d+, \d+, \d+, 1, false'),
       ('`k_igemm` parity classes / row scatter', r'k_igemm<\d+, \d+, \d+, \d+, \d+, [01], true'),
       ('window kernels `k_conv3x3_c64`, `k_wrw3x3_c64`', r'k_(conv|wrw)3x3_c64'),
       ('`k_wrw2` + `k_wrw_reduce`', r'k_wrw(2|_reduce)'),
       ('other 1x1 (`k_conv1x1_stream<*,false>`, `k_conv1x1_fwd`)', r'k_conv1x1_(stream<\d+, false|fwd)'),
       ('`k_bn_bwd_apply`', r'k_bn_bwd_apply'), ('`k_bn_apply`', r'k_bn_apply'),
       ('BN finalize / statistics kernels', r'k_bn_(bwd_)?(finalize|stats)'),
       ('stem + max-pool', r'k_(stem|maxpool)'), ('aten / runtime copies', r'at::|rocblas|__amd')]
busy = sum(float(x[4]) for x in body)
pats = [r'k_conv1x1_stream<\d+, true, ', r'k_igemm<\d+, \d+, \d+, \d+, \d+, 2[,>]', r'k_conv1x1_fwd<\d+, true, ']
t = c = 0
for row in csv.DictReader(open(P('rocprofv3_stats_b256.csv'))):
  if any(re.search(p, row['Name']) for p in pats):
    c += int(row['Calls']); t += float(row['TotalDurationNs'])
mf = open(P('mfma_busy.txt')).read()
mb = re.search(r'= ([\d.]+) % of the matrix pipes busy over the step', mf)
vals = {'VALUE': '{:,.0f}'.format(d['value']).replace(',', ' '), 'MS': '%.2f' % d['ms_per_step'], 'MFMA': '%.1f %%' % (100 * r['frac']),
        'UNSHARED': '%.3f' % h['unshared']['frac'] if h.get('unshared') else 'n/a', 'SHARED': '%.3f' % h['shared']['frac'],
        'TRAFFIC': '%.1f' % (h['traffic'] / 1e6) if h.get('traffic') else 'n/a', 'CPU': '%.1f' % d['cpu_baseline']['value'],
        'C3': '{:,.0f}'.format(others['c3']['value']).replace(',', ' ') if 'c3' in others else 'n/a', 'PASSED': passed}
block = []
block.append('| quantity | value | source |')
block.append('|---|---|---|')
sgc, cal = d['config'].get('step_graph'), d['config'].get('step_mode_calibration')
mode = ('%d replays + %d launch-by-launch steps' % (sgc['replayed_steps'], sgc['launch_by_launch_steps_with_events'])) if sgc else 'launch-by-launch steps'
if cal:
  mode += '; calibration before the timed region: replay %.2f ms, launch by launch %.2f ms' % (cal['replay_ms_per_step'], cal['launch_by_launch_ms_per_step'])
block.append('| images/s (whole job, N = 1) | **%s** (%s ms/step; %s) | `profiles/%s_bench.json` |' % (vals['VALUE'], vals['MS'], mode, tag))
block.append('| `roofline` (headline): step FLOPs / step time / 2.5 PFLOP/s | %.1f TFLOP/s = **%s** of the dense bf16 MFMA peak | same; matrix-pipe busy from counters: %s %% (`%s_mfma_busy.txt`) |' % (
    r['achieved'], vals['MFMA'], mb.group(1) if mb else '?', tag))
if h.get('unshared'):
  block.append('| `hbm_region.unshared`: fused 1x1 forward, 72 launches / step, chip to itself | %.1f us per launch, %.0f GB/s = **%s** of 8 TB/s | same |' % (
      h['unshared']['avg_launch_ms'] * 1e3, h['unshared']['achieved'], vals['UNSHARED']))
block.append('| `hbm_region.shared`: the same launches beside the other stream | %.1f us, %.0f GB/s = %s (student %.1f us, teacher %.1f us) | same |' % (
    h['shared']['avg_launch_ms'] * 1e3, h['shared']['achieved'], vals['SHARED'], h['shared']['by_stream']['main_stream']['avg_launch_ms'] * 1e3,
    h['shared']['by_stream']['teacher_stream']['avg_launch_ms'] * 1e3))
block.append('| the prologue kernels of that region under rocprofv3 | %.1f us over %d launches (the event figure above also holds the teacher\'s prologue-free conv3 launches, which share kernel names with backward-data launches and cannot be told apart in a trace) | `%s_rocprofv3_stats_b256.csv` |' % (t / c / 1e3, c, tag))
if h.get('traffic'):
  block.append('| HBM traffic of the region per launch (2 x FETCH_SIZE + WRITE_SIZE) | %s MB against %.1f MB algorithmic = %.3f x | `%s_pmc_*.csv` |' % (
      vals['TRAFFIC'], h['algorithmic_bytes_per_launch'] / 1e6, h['traffic'] / h['algorithmic_bytes_per_launch'], tag))
cb = d['cpu_baseline']
block.append('| `cpu_baseline` (oracle learner step on the host, baseline only) | %s images/s, %d of %d threads of an %s, B = 32, 5 steps | same |' % (
    vals['CPU'], cb['cores'], cb['cores_available'], cb['cpu_model']))
for cname, what in (('c1', 'C1 ResNet-20 weight sparsification'), ('c2a32', 'C2 with 32-bit activations (reference default)'),
                    ('c3', 'C3 MobileNet-v1 channel-pruned + distillation'), ('c4', 'C4 ResNet-50 NUQ 4-bit + distillation')):
  if cname in others:
    x = others[cname]
    block.append('| %s | %s images/s, %.2f ms/step | `profiles/%s_bench_%s.json` |' % (what, '{:,.0f}'.format(x['value']).replace(',', ' '), x['ms_per_step'], tag, cname))
block.append('| `pytest tests -m gpu` | %s passed, %s skipped (2-GPU RCCL test and friends) | `profiles/%s_pytest_gpu.log` |' % (passed, skipped, tag))
block.append('')
block.append('Where a step goes (`profiles/%s_step_kernels_b256.csv`: %s; kernel time summed over the three queues %.1f ms -- a launch that shares the chip takes longer from start to end, so the sum exceeds the wall time and the backward-filter row, whose launches all run beside backward-data and BN launches, is its largest):' % (
    tag, head[0].strip('# ').replace('"', ''), busy))
block.append('')
block.append('| family | ms / step |')
block.append('|---|---|')
for name, pat in fam:
  block.append('| %s | %.2f |' % (name, tot(pat)))
text = '\n'.join(block)
for rel in ('README.md', os.path.join('profiles', 'README.md'), 'DESIGN.md'):
  p = os.path.join(ROOT, rel)
  s = open(p).read()
  for k, v in vals.items():
    s = s.replace('@@%s@@' % k, v)
  if rel == 'DESIGN.md':
    if 'RESULTS_PLACEHOLDER' in s:
      s = s.replace('RESULTS_PLACEHOLDER', '<!-- results:begin -->\n' + text + '\n<!-- results:end -->')
    else:
      s = re.sub(r'<!-- results:begin -->.*?<!-- results:end -->', lambda m_: '<!-- results:begin -->\n' + text + '\n<!-- results:end -->', s, flags=re.S)
  open(p, 'w').write(s)
print(text)
