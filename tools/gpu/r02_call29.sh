# round 2, GPU call 29: device-memory picture of a bench process (allocator retries / segment churn per step)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 100 python bench.py --steps 6 --warmup 3 --no_cpu_baseline > gpurun_out/r02_c29.log 2>&1
grep -E '"metric"' gpurun_out/r02_c29.log | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print(round(d['value']), round(d['ms_per_step'],2), d['launch_probe']); print(json.dumps(d['memory'], indent=0))"
grep -v '"metric"' gpurun_out/r02_c29.log | grep -E "memory|restart|tottime|empty|run_backward" | head -12 | cut -c1-700
