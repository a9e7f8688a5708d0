#!/usr/bin/env bash
# Device-only gfx950 assembly of every kernel source, one .s per file, with the product's flags (csrc/build.sh):
#   tools/isa_snapshot.sh <out_dir>
# Take one snapshot before and one after a refactoring that must not change the generated code and compare them with
# tools/isa_diff.py (no GPU needed: hipcc cross-compiles).
set -euo pipefail
OUT=${1:?usage: isa_snapshot.sh <out_dir>}
cd "$(dirname "$0")/../pocketflow_amd/csrc"
mkdir -p "$OUT"
FLAGS=$(grep '^FLAGS="--offload-arch' build.sh | sed 's/^FLAGS="//; s/"$//')
PIDS=()
for f in *.hip; do
  /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S "$f" -o "$OUT/${f%.hip}.s" &
  PIDS+=($!)
done
for p in "${PIDS[@]}"; do wait "$p"; done
ls "$OUT" | wc -l
