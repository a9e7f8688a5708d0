#!/usr/bin/env bash
# Build the gfx950 kernel library in-tree:  pocketflow_amd/csrc/libpocketflow_hip.so
# (cross-compiles without a GPU).  -ffp-contract=off: one rounding per float op, like the
# reference's one-TF-op-per-rounding chains; fused multiply-adds are written as fmaf().
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function"
# PF_EXTRA_FLAGS: extra -D switches for the WHOLE product build (empty by default; the scheduling experiments of tools/gpu/build_variant.sh
# become the product by naming them here once a GPU session has measured them)
FLAGS="$FLAGS ${PF_EXTRA_FLAGS:-}"
# ablation / timing macros change what the kernels COMPUTE (tools/gpu/build_ablate.sh builds those as separate libraries): never in the product
case " ${PF_EXTRA_FLAGS:-} " in *_ABLATE*|*_TIMING*) echo "build.sh: refusing ${PF_EXTRA_FLAGS}: ablation / timing builds belong to tools/gpu/build_ablate.sh" >&2; exit 2;; esac
# objects are reused only when they were compiled with the same flags: the flag string is stamped next to them
if [ ! -f .build_flags ] || [ "$(cat .build_flags)" != "$FLAGS" ]; then rm -f ./*.o; printf '%s' "$FLAGS" > .build_flags; fi
OBJS=()
PIDS=()
for f in pf_api pf_quant pf_normalize pf_sparse pf_optim pf_loss pf_bn pf_conv pf_conv_stream pf_igemm pf_conv3x3_c64 pf_wrw3x3_c64 pf_wrw pf_pool pf_transpose pf_stem pf_stem3 pf_image pf_depthwise pf_convg pf_prox pf_im2col; do
  if [ ! -f "$f.o" ] || [ "$f.hip" -nt "$f.o" ] || [ pf_common.h -nt "$f.o" ] || [ pf_conv_common.h -nt "$f.o" ] || [ pf_igemm.h -nt "$f.o" ] || [ ../../include/pocketflow_hip.h -nt "$f.o" ]; then
    rm -f "$f.o"
    $HIPCC $FLAGS -c "$f.hip" -o "$f.o" &
    PIDS+=($!)
  fi
  OBJS+=("$f.o")
done
for p in "${PIDS[@]:-}"; do [ -z "$p" ] || wait "$p" || { echo "build.sh: a compile failed" >&2; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libpocketflow_hip.so "${OBJS[@]}"
echo "built $(pwd)/libpocketflow_hip.so"
