# round 2, GPU call 28: first process on a fresh box; if it comes up host-bound, bench.py profiles one step of it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python bench.py --steps 6 --warmup 3 --no_cpu_baseline > gpurun_out/r02_c28.log 2>&1
grep -E '"metric"' gpurun_out/r02_c28.log | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print(round(d['value']), round(d['ms_per_step'],2), d['launch_probe'])"
grep -v '"metric"' gpurun_out/r02_c28.log | head -60 | cut -c1-200
