# Round 4, last GPU call: the recorded-step tests after the hand-over fix, bench.py as the driver runs it (plain with cpu_baseline, and
# under rocprofv3 --stats), one line per other configuration.
TAG=r04
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_learner_gpu.py -m gpu -q --timeout=900 --tb=short 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-400 | tee gpurun_out/${TAG}_pytest_gpu_learners_last.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/${TAG}_prof.log > gpurun_out/${TAG}_bench_under_rocprof.json
python tools/prof_summary.py $(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/${TAG}_step_kernels_b256.csv | head -4 | cut -c1-150
cp $(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_rocprofv3_stats_b256.csv
timeout 900 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; grep '"metric"' gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
for c in c2a32 c4 c3 c1; do
  timeout 400 python bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err || tail -3 gpurun_out/${TAG}_bench_$c.err
done
python - <<'PY'
import json, csv, re
for name in ('r04_bench_under_rocprof', 'r04_bench', 'r04_bench_c2a32', 'r04_bench_c4', 'r04_bench_c3', 'r04_bench_c1'):
  try:
    d = json.loads([l for l in open('gpurun_out/%s.json' % name) if l.startswith('{')][0])
    r = d['roofline']
    u = r.get('unshared')
    print(name, round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms | host median', round(d['host_submit_ms_min_median_max'][1], 2), '| region frac', round(r['frac'], 4), 'avg us', round(1e3 * (r['avg_launch_ms'] or 0), 1), 'n', r['launches'], '| unshared', u and (round(u['frac'], 4), round(1e3 * u['avg_launch_ms'], 1), u['launches']))
  except Exception as e:
    print(name, 'failed', e)
rows = list(csv.DictReader(open('gpurun_out/r04_rocprofv3_stats_b256.csv')))
pats = [r'k_conv1x1_stream<\d+, true, ', r'k_igemm<\d+, \d+, \d+, \d+, \d+, 2[,>]', r'k_conv1x1_fwd<\d+, true, ']
tot = n = 0
for r in rows:
  if any(re.search(p, r['Name']) for p in pats):
    tot += float(r['TotalDurationNs']); n += int(r['Calls'])
print('rocprofv3 stats: region launches', n, 'avg us %.1f' % (tot / n / 1e3), 'frac %.4f' % (312.92e6 / (tot / n * 1e-9) / 8e12))
PY
