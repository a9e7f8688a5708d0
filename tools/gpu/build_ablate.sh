#!/usr/bin/env bash
# ablation builds of pf_igemm.hip for tools/gpu/igemm_ablate.py (tools only; never linked into the product library)
set -euo pipefail
cd "$(dirname "$0")/../../pocketflow_amd/csrc"
mkdir -p ../../tools/gpu/_build
for n in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wno-unused-function -DPF_IG_ABLATE=$n -shared pf_igemm.hip -o ../../tools/gpu/_build/libig_ablate$n.so &
done
wait
for n in 1 2 4 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wno-unused-function -DPF_W2_ABLATE=$n -shared pf_wrw.hip -o ../../tools/gpu/_build/libwrw_ablate$n.so &
done
wait
