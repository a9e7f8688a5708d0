# Round 6, GPU calls 15 (several, each a short diagnostic of the masked MobileNet fine-tune parity failure; last content: which of the
# round's changes moved the bf16 gradient checks of the two small networks -- the backward-filter queue or the BN finalize order)
# (libpocketflow_hip_oldfin.so: tools/gpu/build_variant.sh oldfin with pf_bn.hip of commit 6b05ab2 -- the k_bn_finalize of rounds 1-5 -- in place of the tree's;
#  built by hand for this round's A/B calls, tools/gpu/_build/ is not tracked)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "--- one queue"
PF_WRW_SIDE=0 timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=line -k "ws_resnet20_bf16 or cp_mobilenet_bf16" 2>&1 | grep -E "passed|failed|Error" | cut -c1-700
echo "--- old finalize, side queue"
PF_HIP_LIB=$PWD/tools/gpu/_build/libpocketflow_hip_oldfin.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=line -k "ws_resnet20_bf16 or cp_mobilenet_bf16" 2>&1 | grep -E "passed|failed|Error" | cut -c1-700
echo "--- old finalize, one queue"
PF_WRW_SIDE=0 PF_HIP_LIB=$PWD/tools/gpu/_build/libpocketflow_hip_oldfin.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=line -k "ws_resnet20_bf16 or cp_mobilenet_bf16" 2>&1 | grep -E "passed|failed|Error" | cut -c1-700
exit 0
