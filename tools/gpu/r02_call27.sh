# round 2, GPU call 27: bench.py with the host-share check (single process and two ranks on one GPU)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python bench.py --steps 6 --warmup 3 --no_cpu_baseline 2>&1 | grep -E '"metric"|restarting|Error|error' | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(round(d['value']), round(d['ms_per_step'],2), d['launch_probe'])
    else: print(ln.strip()[:200])"
timeout 600 python -m pytest tests/test_learner_gpu.py -q --tb=short -k "two_ranks" 2>&1 | tail -3 | cut -c1-300
