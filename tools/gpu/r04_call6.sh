# Round 4, sixth GPU call: (1) is MIOpen's naive solver what kills hipStreamEndCapture in the tests (env now set by the package)?
# (2) where did the "own" configuration lose 6 ms (kernel trace), after the SGPR fix of k_igemm; (3) bench A/B of each own piece;
# (4) the proximal-step kernels.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { tag=$1; shift; timeout 1200 "$@" > gpurun_out/r04_c6_$tag.log 2>&1; echo "== $tag rc=$?"; grep -v "amdgpu.ids\|Warning\|warnings.warn" gpurun_out/r04_c6_$tag.log | tail -${TAILN:-6} | cut -c1-330; }
TAILN=8 run step_graph python -X faulthandler -m pytest tests/test_learner_gpu.py -m gpu -q --tb=short -k step_graph -s
run prox python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "proximal"
line() { python -c "
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d = json.loads(ln); print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms/step | roofline frac', round(d['roofline']['frac'], 4), '| host', [round(v, 1) for v in d['host_submit_ms_min_median_max']])
" $1 "$2"; }
i=0
for v in "X=1" "PF_OWN_CONV_GENERIC=0" "PF_OWN_CONV2D_BWD_STRIDED=0" "PF_OWN_CONV2D_WRW_MIN_C=64" "PF_OWN_CONV_GENERIC=0 PF_OWN_CONV2D_BWD_STRIDED=0" "X=2"; do
  i=$((i+1))
  env $v timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r04_c6_bench_$i.json 2> gpurun_out/r04_c6_bench_$i.err
  line gpurun_out/r04_c6_bench_$i.json "$v"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c6 -o c6 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 5 --no_cpu_baseline --step_graph 0 > $GRAFT_REPO_ROOT/gpurun_out/r04_c6_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find /tmp/prof_c6 -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r04_c6_step_kernels.csv | head -30 | cut -c1-170
