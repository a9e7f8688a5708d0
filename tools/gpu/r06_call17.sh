# Round 6, GPU call 17: graph.WrwSide with the data-parallel reducer (two ranks on one GPU over gloo), the learner / parity suites behind it,
# bench.py with the step-mode calibration
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 2400 python -m pytest tests/test_learner_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -8 | cut -c1-400
for i in 1 2; do timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('%.0f images/s  %.2f ms/step  calibration %s  filter queue %s' % (d['value'], d['ms_per_step'], d['config']['step_mode_calibration'], d['config']['backward_filter_queue']))
"; done
tail -3 $O/r06_ab_err.txt
exit 0
