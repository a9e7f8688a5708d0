#!/usr/bin/env bash
# First GPU call of round 5 (prepared at the end of round 4, when no GPU minutes were left):
#   gpurun --timeout 2400 -- 'bash tools/gpu/r05_first_call.sh'        (typically ~20 min: the suite ~10, the rest ~10)
# 1. the whole GPU suite at HEAD (validates the end-of-round-4 clean-up on hardware: kernel machine code is unchanged per
#    tools/isa_diff.py, the host dispatch lost two dead branches);
# 2. the prepared, never-run pieces: PF_DW_REDUCE2 and PF_CONVG_PAD_C3 (their tests, then C3 with and without them);
# 3. the LDS-DMA fill-rate table (tools/gpu/fill_bench.hip) that decides what the next contraction kernel should look like;
# 4. the headline line.
# Everything lands in gpurun_out/r05_first_call/ ; delete this script after use (it stays in the history).
set -uo pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_first_call
mkdir -p "$OUT"
export TMPDIR=/tmp
( time timeout 1000 python -m pytest tests -m gpu -x -q ) > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"
PF_TEST_PREPARED=1 timeout 240 python -m pytest tests/test_depthwise_gpu.py -m gpu -q > "$OUT/pytest_prepared_dw_reduce2.log" 2>&1
tail -2 "$OUT/pytest_prepared_dw_reduce2.log"
PF_TEST_PREPARED=1 timeout 240 python -m pytest tests/test_igemm_gpu.py -m gpu -q -k "256x256" > "$OUT/pytest_prepared_tile256x256.log" 2>&1
PF_TEST_PREPARED=1 timeout 300 python -m pytest tests/test_conv_gpu.py -m gpu -q -k "prologue_three_stage" > "$OUT/pytest_prepared_pro256.log" 2>&1
tail -2 "$OUT/pytest_prepared_pro256.log"
tail -2 "$OUT/pytest_prepared_tile256x256.log"
timeout 300 bash tools/gpu/fill_bench.sh > /dev/null 2>&1; cp gpurun_out/fill_bench.txt "$OUT/fill_bench.txt" 2>/dev/null
# PF_CONVG_PAD_C3=1 (image convolutions on k_convg's vector loader): the parity tests of the networks that have one, then C3 with both switches
PF_CONVG_PAD_C3=1 timeout 400 python -m pytest tests/test_parity_gpu.py tests/test_convg_gpu.py -m gpu -q -k "mobilenet or lenet or cp or convg" > "$OUT/pytest_prepared_pad_c3.log" 2>&1
tail -2 "$OUT/pytest_prepared_pad_c3.log"
for v in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do
  set -- $v
  echo "PF_DW_REDUCE2=$1 PF_CONVG_PAD_C3=$2" >> "$OUT/c3_switches_ab.txt"
  PF_DW_REDUCE2=$1 PF_CONVG_PAD_C3=$2 timeout 200 python bench.py --config c3 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1 >> "$OUT/c3_switches_ab.txt"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1 > "$OUT/bench_c2.json"
# PF_IGEMM_AUTO256 / PF_IGEMM_PRO256 (256 x 256 tiles where N % 256 == 0 and >= T tiles; plain / prologue launches): only meaningful if
# the two prepared tile tests above passed
for v in "0 0" "128 0" "0 128" "128 128" "0 0" "128 128" "64 256"; do
  set -- $v
  echo "PF_IGEMM_AUTO256=$1 PF_IGEMM_PRO256=$2" >> "$OUT/c2_auto256_ab.txt"
  PF_IGEMM_AUTO256=$1 PF_IGEMM_PRO256=$2 timeout 200 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | tail -1 >> "$OUT/c2_auto256_ab.txt"
done
timeout 400 python tools/gpu/igemm_bench.py > "$OUT/igemm_layers.txt" 2>&1   # (now with the 256x256 tile column)
timeout 300 python tools/gpu/depthwise_bench.py > "$OUT/depthwise_layers.txt" 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05_first_call/*.txt')) + ['gpurun_out/r05_first_call/bench_c2.json']:
  for line in open(f):
    line = line.strip()
    if line.startswith('{'):
      try:
        d = json.loads(line); print(f.split('/')[-1], d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'))
      except Exception as e:
        print(f, 'unparsable', e)
    elif line.startswith('PF_'):
      print(line)
PY
