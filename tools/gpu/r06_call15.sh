cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -k "cp_mobilenet_masked" 2>&1 | tail -8 | cut -c1-400
PF_HIP_LIB=$PWD/tools/gpu/_build/libpocketflow_hip_oldfin.so timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -k "cp_mobilenet_masked" 2>&1 | tail -4 | cut -c1-400
exit 0
