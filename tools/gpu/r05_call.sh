#!/usr/bin/env bash
set -uo pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_c3c
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_igemm_gpu.py -m gpu -q -x -k "stem3" > "$OUT/pytest_stem3.log" 2>&1; tail -4 "$OUT/pytest_stem3.log"
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_conv_gpu.py -m gpu -q -x -k "pool" > "$OUT/pytest_pool.log" 2>&1; tail -2 "$OUT/pytest_pool.log"
for i in 1 2; do timeout 200 python bench.py --config c3 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1 >> "$OUT/c3.txt"; done
for v in old new old new; do
  echo "maxpool=$v" >> "$OUT/c2_pool_ab.txt"
  if [ $v = old ]; then export PF_HIP_LIB=$PWD/tools/gpu/_build/libpocketflow_hip_oldpool.so; else unset PF_HIP_LIB; fi
  timeout 200 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1 >> "$OUT/c2_pool_ab.txt"
done
unset PF_HIP_LIB
python - <<'PY'
import json
for l in open('gpurun_out/r05_c3c/c3.txt'):
  if l.startswith('{'):
    d=json.loads(l); print('c3', round(d['value']), round(d['ms_per_step'],2))
lab=None
for l in open('gpurun_out/r05_c3c/c2_pool_ab.txt'):
  l=l.strip()
  if l.startswith('maxpool'): lab=l
  elif l.startswith('{'):
    d=json.loads(l); print(lab, round(d['value']), round(d['ms_per_step'],2))
PY
