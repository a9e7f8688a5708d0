# Round 4, call 11: re-sweep of the tile / split choices after the scheduling changes (the defaults were measured in rounds 2-3).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python tools/gpu/igemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_igemm_layers.txt | cut -c1-230
TARGETS=192,256,320,384,512 timeout 400 python tools/gpu/wrw_target_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_wrw_target_bench.txt | cut -c1-200
timeout 300 python tools/gpu/fwd1x1_layers.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_fwd1x1_layers.txt | cut -c1-200
