# the bench command plain and under rocprofv3 --stats: does roofline (both streams' region launches) agree with the trace's average?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r04_last_bench.json 2> gpurun_out/r04_last_bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_last -o last -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/r04_last_prof.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/r04_last_prof.log > gpurun_out/r04_last_bench_under_rocprof.json
cp $(find /tmp/prof_last -name '*kernel_stats.csv' | head -1) gpurun_out/r04_last_stats.csv
python tools/prof_summary.py $(find /tmp/prof_last -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r04_last_step_kernels.csv | head -3 | cut -c1-150
python - <<'PY'
import json, csv, re
for name in ('r04_last_bench', 'r04_last_bench_under_rocprof'):
  d = json.loads([l for l in open('gpurun_out/%s.json' % name) if l.startswith('{')][0])
  r = d['roofline']
  print(name, round(d['value']), 'img/s | region frac', round(r['frac'], 4), 'avg us', round(1e3 * r['avg_launch_ms'], 1), 'n', r['launches'], r['by_stream'])
rows = list(csv.DictReader(open('gpurun_out/r04_last_stats.csv')))
pats = [r'k_conv1x1_stream<\d+, true, ', r'k_igemm<\d+, \d+, \d+, \d+, \d+, 2[,>]', r'k_conv1x1_fwd<\d+, true, ']
tot = n = 0
for r in rows:
  if any(re.search(p, r['Name']) for p in pats):
    tot += float(r['TotalDurationNs']); n += int(r['Calls'])
print('rocprofv3 stats: region launches', n, 'avg us %.1f' % (tot / n / 1e3), 'frac %.4f' % (312.92e6 / (tot / n * 1e-9) / 8e12))
PY
