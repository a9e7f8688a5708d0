# Round 4, eighth GPU call: bisect the hipStreamEndCapture crash of the recorded-step worker (bench.py records the same step fine).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
w() { tag=$1; shift; rm -rf /tmp/w_$tag; mkdir -p /tmp/w_$tag; env "$@" timeout 300 python tests/step_graph_worker.py uq_resnet50 /tmp/w_$tag > gpurun_out/r04_c8_$tag.log 2>&1; echo "worker $tag [$*] rc=$? $(grep -o 'STEP_GRAPH_RESULT.\{0,160\}' gpurun_out/r04_c8_$tag.log | cut -c1-200)"; }
w base X=1
w keepout PF_W_DROP_OUT=0
w pool2 PF_W_POOL=2
w bench PF_W_BENCHMARK=1
w nostrict PF_W_STRICT=0
w noahead PF_TEACHER_AHEAD=0
w all PF_W_POOL=2 PF_W_BENCHMARK=1 PF_W_STRICT=0
timeout 300 python bench.py --config c2 --batch 8 --image_size 64 --steps 6 --warmup 5 --no_cpu_baseline > gpurun_out/r04_c8_bench_small.log 2>&1; echo "bench small rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/r04_c8_bench_small.log)"
line() { python -c "
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d = json.loads(ln); print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms/step | roofline frac', round(d['roofline']['frac'], 4), '| host', [round(v, 1) for v in d['host_submit_ms_min_median_max']])
" $1 "$2"; }
for c in c2 c1 c3 c4; do
  timeout 400 python bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r04_c8_bench_$c.json 2> gpurun_out/r04_c8_bench_$c.err; line gpurun_out/r04_c8_bench_$c.json "$c"
done
PF_TEACHER_AHEAD=0 timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r04_c8_bench_c2_noahead.json 2>/dev/null; line gpurun_out/r04_c8_bench_c2_noahead.json "c2 teacher in line"
timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --step_graph 0 > gpurun_out/r04_c8_bench_c2_eager.json 2>/dev/null; line gpurun_out/r04_c8_bench_c2_eager.json "c2 launch by launch"
