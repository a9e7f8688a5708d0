cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "--- one queue"
PF_WRW_SIDE=0 timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=line -k "ws_resnet20_bf16 or cp_mobilenet_bf16" 2>&1 | grep -E "passed|failed|Error" | cut -c1-700
echo "--- old finalize, side queue"
PF_HIP_LIB=$PWD/tools/gpu/_build/libpocketflow_hip_oldfin.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=line -k "ws_resnet20_bf16 or cp_mobilenet_bf16" 2>&1 | grep -E "passed|failed|Error" | cut -c1-700
echo "--- old finalize, one queue"
PF_WRW_SIDE=0 PF_HIP_LIB=$PWD/tools/gpu/_build/libpocketflow_hip_oldfin.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=line -k "ws_resnet20_bf16 or cp_mobilenet_bf16" 2>&1 | grep -E "passed|failed|Error" | cut -c1-700
exit 0
