#!/usr/bin/env bash
# ablation builds of pf_igemm.hip / pf_wrw.hip for tools/gpu/*_ablate.py (tools only; never linked into the product library).
# The builds need symbols of other files of the product library (pf_wrw_reduce, the tuning switches): they LINK against it (their own kernels come first in their
# local lookup scope).  Loading the product library RTLD_GLOBAL instead makes the dynamic linker bind the weak template
# kernel stubs of every ablation build to the PRODUCT's kernels -- round 3 measured the same kernel five times that way.
set -euo pipefail
cd "$(dirname "$0")/../../pocketflow_amd/csrc"
mkdir -p ../../tools/gpu/_build
for n in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wno-unused-function -Wno-unused-variable -DPF_IG_ABLATE=$n -shared pf_igemm.hip -o ../../tools/gpu/_build/libig_ablate$n.so -L. -l:libpocketflow_hip.so -Wl,-rpath,'$ORIGIN/../../../pocketflow_amd/csrc' &
done
wait
for n in 1 2 4 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wno-unused-function -DPF_W2_ABLATE=$n -shared pf_wrw.hip -o ../../tools/gpu/_build/libwrw_ablate$n.so -L. -l:libpocketflow_hip.so -Wl,-rpath,'$ORIGIN/../../../pocketflow_amd/csrc' &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -Wno-unused-function -DPF_W2_TIMING -shared pf_wrw.hip -o ../../tools/gpu/_build/libwrw_timing.so -L. -l:libpocketflow_hip.so -Wl,-rpath,'$ORIGIN/../../../pocketflow_amd/csrc'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -Wno-unused-function -DPF_IG_TIMING -shared pf_igemm.hip -o ../../tools/gpu/_build/libig_timing.so -L. -l:libpocketflow_hip.so -Wl,-rpath,'$ORIGIN/../../../pocketflow_amd/csrc'
