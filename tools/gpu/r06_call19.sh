# Round 6, GPU call 19: k_bn_finalize in the summation order of rounds 1-5 (bit for bit against the old kernel, kept as a variant
# library), the gradient checks that moved with the order, then the round's evidence run
# (libpocketflow_hip_oldfin.so: tools/gpu/build_variant.sh oldfin with pf_bn.hip of commit 6b05ab2 -- the k_bn_finalize of rounds 1-5 -- in place of the tree's;
#  built by hand for this round's A/B calls, tools/gpu/_build/ is not tracked)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/gpu/bn_finalize_ab.py tools/gpu/_build/libpocketflow_hip_oldfin.so pocketflow_amd/csrc/libpocketflow_hip.so 2>&1 | tail -4 | cut -c1-200 | tee gpurun_out/r06_bn_finalize_order.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=line -k "ws_resnet20_bf16 or cp_mobilenet_bf16 or one_step_at_224 or cp_mobilenet_masked" 2>&1 | grep -E "passed|failed|Error" | cut -c1-500 | tee -a gpurun_out/r06_bn_finalize_order.txt
if grep -q failed gpurun_out/r06_bn_finalize_order.txt; then exit 0; fi
bash tools/gpu/round_evidence.sh r06
