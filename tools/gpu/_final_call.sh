cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_final
for rep in 1 2; do
for ahead in 0 1; do
  for pp in 0 1; do
    PF_TEACHER_AHEAD=$ahead PF_IGEMM_PP=$pp timeout 200 python bench.py --no_cpu_baseline --unshared_steps 0 > /tmp/b.json 2>/tmp/b.err
    python - <<PY
import json
for ln in open('/tmp/b.json'):
    if ln.startswith('{'):
        d = json.loads(ln); print('teacher branch %s  PF_IGEMM_PP=%s  %.0f images/s  %.2f ms/step' % ('beside the forward pass' if $ahead else 'in line (one stream)   ', $pp, d['value'], d['ms_per_step']))
PY
  done
done
done | tee gpurun_out/r05_final/pp_teacher_ab.txt
