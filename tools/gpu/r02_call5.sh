# round 2, GPU call 5: 256x256 igemm tiles, PMC view of the igemm / fused / wrw kernels, headline bench after the reverts
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/gpu/igemm_bench.py > gpurun_out/r02_c5_igemm_bench.log 2>&1; tail -14 gpurun_out/r02_c5_igemm_bench.log | cut -c1-230
PF_IGEMM_TILE=128x128 bash tools/gpu/pmc_kernel.sh ig128 igemm 14 256 256 3 1
PF_IGEMM_TILE=256x256 bash tools/gpu/pmc_kernel.sh ig256 igemm 14 256 256 3 1
bash tools/gpu/pmc_kernel.sh fused14 fused 14 1024 256 0
bash tools/gpu/pmc_kernel.sh fused56 fused 56 64 256 1
bash tools/gpu/pmc_kernel.sh wrw14 wrw 14 1024 256
for f in gpurun_out/pmc_*.csv; do echo "== $f"; cat $f | cut -c1-250; done
timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c5_bench.log 2>&1; tail -1 gpurun_out/r02_c5_bench.log | cut -c1-400
timeout 600 python -m pytest tests/test_igemm_gpu.py tests/test_conv_gpu.py -q --tb=short -k "wrw" 2>&1 | tail -6 | cut -c1-300
