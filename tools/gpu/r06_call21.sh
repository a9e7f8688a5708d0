# Round 6, GPU call 21: depthwise / im2col / MobileNet-stem backward-filter launches on the side queue too -- recorded (no forks) against
# launch-by-launch (forks) runs of the three learner kinds, the small-network parity checks, C3 / C1 / C2 lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests/test_learner_gpu.py -m gpu -q -s --tb=short -k "side_queue or step_graph_is_the_eager" 2>&1 | grep -E "step graph|passed|failed|Error" | cut -c1-300
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=line -k "ws_resnet20 or cp_mobilenet" 2>&1 | grep -E "passed|failed|Error" | cut -c1-400
for c in c3 c3 c1 c2; do
timeout 400 python bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); c = d['config']['step_mode_calibration']; print('$c %.0f images/s  %.2f ms/step  replay %.2f lbl %.2f kept %s' % (d['value'], d['ms_per_step'], c['replay_ms_per_step'], c['launch_by_launch_ms_per_step'], c['kept']))
"; done
tail -2 $O/r06_ab_err.txt
exit 0
