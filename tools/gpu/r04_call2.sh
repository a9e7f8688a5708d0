# Round 4, second GPU call: the promoted scheduling build + teacher-ahead default + the recorded step (step_graph.py).
#   gpurun --timeout 1500 -- 'bash tools/gpu/r04_call2.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
line() { python -c "
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d = json.loads(ln); print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms/step | host submit', [round(v, 2) for v in d.get('host_submit_ms_min_median_max')], '| roofline frac', round(d['roofline']['frac'], 4), '| region ms per launch', d['roofline'].get('avg_launch_ms'), '|', d['config'].get('step_graph'), d['config'].get('teacher'))
" $1 $2; }
# 1. the recorded step must BE the eager step (bit for bit), three learners
PF_TEST_POISON=1 timeout 900 python -m pytest tests/test_learner_gpu.py -m gpu -q -x --tb=short -k "step_graph" -s 2>&1 | grep -v amdgpu.ids | tail -25 | cut -c1-400 | tee gpurun_out/r04_c2_pytest_step_graph.log
# 2. the step: recorded vs launch by launch, c2 / c1 / c3 / c4
for cfg in c2 c1 c3 c4; do
  for sg in 1 0; do
    timeout 400 python bench.py --config $cfg --steps 20 --warmup 5 --no_cpu_baseline --step_graph $sg > gpurun_out/r04_c2_bench_${cfg}_sg$sg.json 2> gpurun_out/r04_c2_bench_${cfg}_sg$sg.err || tail -5 gpurun_out/r04_c2_bench_${cfg}_sg$sg.err
    line gpurun_out/r04_c2_bench_${cfg}_sg$sg.json ${cfg}-graph$sg
    grep -i "step graph\|not recorded" gpurun_out/r04_c2_bench_${cfg}_sg$sg.err | head -3 | cut -c1-300
  done
done
PF_TEACHER_AHEAD=0 timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --step_graph 0 > gpurun_out/r04_c2_bench_c2_noahead.json 2> gpurun_out/r04_c2_bench_c2_noahead.err; line gpurun_out/r04_c2_bench_c2_noahead.json c2-eager-teacher-in-line
# 3. kernel tests that touch the tuning-switch plumbing (one-time read + reload) and the promoted scheduling build
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py tests/test_kernels_gpu.py -m gpu -q -x --tb=line 2>&1 | tail -4 | cut -c1-300 | tee gpurun_out/r04_c2_pytest_kernels.log
