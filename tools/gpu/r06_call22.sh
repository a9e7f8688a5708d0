# Round 6, GPU call 22: bench.py after its last edit (one GPU, default line without the CPU baseline; two ranks on one GPU through the test)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('%.0f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], d['config']['step_mode_calibration']))
"
timeout 600 python -m pytest tests/test_learner_gpu.py -m gpu -q --tb=short -k "two_ranks_share" 2>&1 | tail -2 | cut -c1-200
exit 0
