#!/usr/bin/env bash
# Build and run tools/gpu/fill_bench.hip on the GPU box (LDS-DMA fill rate of a CU by working set / access pattern / depth);
# the table goes to gpurun_out/fill_bench.txt -> copy it to profiles/<round>_fill_bench.txt.
#   gpurun --timeout 300 -- 'bash tools/gpu/fill_bench.sh'
set -euo pipefail
cd "$(dirname "$0")/../.."
mkdir -p tools/gpu/_build gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -o tools/gpu/_build/fill_bench tools/gpu/fill_bench.hip
timeout 240 tools/gpu/_build/fill_bench | tee gpurun_out/fill_bench.txt
