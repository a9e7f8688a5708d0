# Round 4, seventh GPU call: recorded-step cases in their own process; convg with split contraction; clean A/B of the own pieces and of
# the side-stream backward-filter launches (bench now re-grows the eager pool after recording).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { tag=$1; shift; timeout 1200 "$@" > gpurun_out/r04_c7_$tag.log 2>&1; echo "== $tag rc=$?"; grep -v "amdgpu.ids\|Warning\|warnings.warn" gpurun_out/r04_c7_$tag.log | tail -${TAILN:-6} | cut -c1-330; }
TAILN=10 run step_graph python -m pytest tests/test_learner_gpu.py -m gpu -q --tb=short -k step_graph -s
run convg python -m pytest tests/test_convg_gpu.py -m gpu -q -x --tb=short
line() { python -c "
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d = json.loads(ln); print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms/step | roofline frac', round(d['roofline']['frac'], 4), '| host', [round(v, 1) for v in d['host_submit_ms_min_median_max']])
" $1 "$2"; }
i=0
for v in "X=warm" "X=1" "PF_WRW_SIDE=0" "PF_OWN_CONV_GENERIC=0" "PF_OWN_CONV2D_BWD_STRIDED=0" "PF_OWN_CONV2D_WRW_MIN_C=64" "PF_TEACHER_AHEAD=0" "X=2" "PF_WRW_SIDE=0"; do
  i=$((i+1))
  env $v timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r04_c7_bench_$i.json 2> gpurun_out/r04_c7_bench_$i.err
  line gpurun_out/r04_c7_bench_$i.json "$v"
done
for c in c1 c3; do
  for v in "X=1" "PF_WRW_SIDE=0"; do
    env $v timeout 400 python bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r04_c7_bench_${c}_$v.json 2> gpurun_out/r04_c7_bench_${c}.err
    line gpurun_out/r04_c7_bench_${c}_$v.json "$c $v"
  done
done
