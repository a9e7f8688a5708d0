# Round 6, GPU call 20: the bench lines of the round with the final bench.py (launch-by-launch mode: backlog built up untimed)
TAG=r06
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
PF_BENCH_TRACE_STEPS=1 timeout 900 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; grep -v '"metric"' gpurun_out/${TAG}_bench.log | tail -2 | cut -c1-300; grep '"metric"' gpurun_out/${TAG}_bench.log | cut -c1-1500
grep '"metric"' gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
for c in c2a32 c4 c3 c1; do
  timeout 400 python bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err || tail -3 gpurun_out/${TAG}_bench_$c.err
  python -c "
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d = json.loads(ln); print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms/step', d['config'].get('step_mode_calibration'), d['host_submit_ms_min_median_max'], d['memory']['after_warmup']['segments_allocated'], d['memory']['after_timed_region']['segments_allocated'])
" gpurun_out/${TAG}_bench_$c.json $c
done
timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --step_graph 1 2>/dev/null | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('c2 --step_graph 1 (recorded step forced): %.0f images/s  %.2f ms/step' % (d['value'], d['ms_per_step']))
" | tee gpurun_out/${TAG}_recorded_forced.txt
exit 0
