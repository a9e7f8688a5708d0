# Round 4, ninth GPU call: the whole GPU suite at HEAD + one bench line per configuration.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r04_parity_report.txt
PF_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r04_parity_report.txt timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --tb=short -x 2>&1 | grep -v "amdgpu.ids" | tail -25 | cut -c1-300 > gpurun_out/r04_pytest_gpu.log
cat gpurun_out/r04_pytest_gpu.log | tail -14
line() { python -c "
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d = json.loads(ln); print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms/step | roofline frac', round(d['roofline']['frac'], 4), '| host', [round(v, 1) for v in d['host_submit_ms_min_median_max']], d['host_submit_ms_steps'][-4:])
" $1 "$2"; }
for c in c2 c1 c3 c4; do
  timeout 400 python bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r04_c9_bench_$c.json 2> gpurun_out/r04_c9_bench_$c.err; line gpurun_out/r04_c9_bench_$c.json "$c"
done
PF_OWN_CONV_IM2COL=0 timeout 400 python bench.py --config c1 --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r04_c9_bench_c1_miopen.json 2>/dev/null; line gpurun_out/r04_c9_bench_c1_miopen.json "c1 few-channel convolutions on MIOpen"
timeout 400 python bench.py --config c1 --steps 20 --warmup 5 --no_cpu_baseline --event_steps 0 > gpurun_out/r04_c9_bench_c1_ev0.json 2>/dev/null; line gpurun_out/r04_c9_bench_c1_ev0.json "c1 no launch-by-launch steps"
