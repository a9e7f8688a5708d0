# Round 6, GPU call 18: graph.WrwSide keeps the producer BN's constants referenced until the join (they were freed under the side stream's
# reads: bf16 gradient checks of ResNet-20 / MobileNet failed in the evidence run); no forks inside a recording; gc.freeze in bench.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -k "ws_resnet20_bf16 or cp_mobilenet_bf16 or one_step_at_224 or cp_mobilenet_masked or nuq_resnet50_4bit_bf16" 2>&1 | tail -12 | cut -c1-500
timeout 1500 python -m pytest tests/test_learner_gpu.py -m gpu -q --tb=short -k "side_queue or step_graph_is_the_eager or two_ranks_share" 2>&1 | tail -5 | cut -c1-300
for c in c2 c3 c1; do
for i in 1 2; do timeout 400 python bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); c = d['config']['step_mode_calibration']; print('$c %.0f images/s  %.2f ms/step  replay %.2f lbl %.2f kept %s; host %s' % (d['value'], d['ms_per_step'], c['replay_ms_per_step'], c['launch_by_launch_ms_per_step'], c['kept'], d['host_submit_ms_min_median_max']))
"; done; done
PF_BENCH_MAIN_PRIORITY=1 timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('c2, student queue at high priority: %.0f images/s  %.2f ms/step' % (d['value'], d['ms_per_step']))
"
tail -3 $O/r06_ab_err.txt
exit 0
