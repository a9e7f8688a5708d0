#!/usr/bin/env bash
# round 5: ping-pong kernel v2 (straight MFMA blocks per fragment-row count, interleaved reads + LDS-DMA, counted waits)
set -uo pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_pp3
mkdir -p "$OUT"
export TMPDIR=/tmp
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
timeout 600 python -m pytest tests/test_igemm_gpu.py -m gpu -q -x -k "pingpong" > "$OUT/pytest_pp.log" 2>&1
tail -5 "$OUT/pytest_pp.log"; stamp tests
timeout 400 python tools/gpu/pp_bench.py 2>&1 | grep -v amdgpu.ids > "$OUT/pp_bench.txt"
cat "$OUT/pp_bench.txt"; stamp pp_bench
PF_HIP_LIB=$PWD/tools/gpu/_build/libpocketflow_hip_pptime.so timeout 200 python tools/gpu/pp_timeline.py 2>&1 | grep -v amdgpu.ids > "$OUT/pp_timeline.txt"
head -45 "$OUT/pp_timeline.txt"; stamp timeline
SH="28,128,128,3,1;14,256,256,3,1;7,512,512,3,1;14,1024,256,1,1"
echo "== PP_ABLATE=3" >> "$OUT/pp_ablate.txt"
PF_HIP_LIB=$PWD/tools/gpu/_build/libpocketflow_hip_ppab3.so PP_SHAPES="$SH" PP_BMS="auto,256" timeout 200 python tools/gpu/pp_bench.py 2>&1 | grep -v amdgpu.ids >> "$OUT/pp_ablate.txt"
cat "$OUT/pp_ablate.txt"; stamp ablate
