#!/usr/bin/env bash
# A COMPLETE variant of the kernel library with extra compiler flags, for A/B runs of the test suite and of bench.py in one box:
#   bash tools/gpu/build_variant.sh x -DPF_SOME_EXPERIMENT   ->  tools/gpu/_build/libpocketflow_hip_x.so
#   PF_HIP_LIB=$PWD/tools/gpu/_build/libpocketflow_hip_x.so python bench.py --no_cpu_baseline
# (round 4 measured the sched_group_barrier pipelines this way before they became the product: profiles/r04_first_call_ab.txt)
# (pocketflow_amd/hip.py loads $PF_HIP_LIB instead of the in-tree library when it is set; tools only.)
set -euo pipefail
name=$1; shift
cd "$(dirname "$0")/../../pocketflow_amd/csrc"
out=../../tools/gpu/_build/variant_$name
mkdir -p $out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function $*"
OBJS=()
for f in pf_api pf_quant pf_normalize pf_sparse pf_optim pf_loss pf_bn pf_conv pf_conv_stream pf_igemm pf_conv3x3_c64 pf_wrw3x3_c64 pf_wrw pf_pool pf_transpose pf_stem pf_stem3 pf_image pf_depthwise pf_convg pf_prox pf_im2col; do
  /opt/rocm/bin/hipcc $FLAGS -c "$f.hip" -o "$out/$f.o" &
  OBJS+=("$out/$f.o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/gpu/_build/libpocketflow_hip_$name.so "${OBJS[@]}"
echo "built tools/gpu/_build/libpocketflow_hip_$name.so"
