# Round 4, tenth GPU call: the test files behind the one that stopped call 9 (-x), per-layer numbers of the new kernels.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_profile_tools.py tests/test_repo_rules.py tests/test_zz_search_gpu.py tests/test_convg_gpu.py -m gpu -q --timeout=900 --tb=short 2>&1 | grep -v "amdgpu.ids" | tail -12 | cut -c1-300
timeout 300 python tools/gpu/new_kernels_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_new_kernels_layers.txt | cut -c1-200
