# round 2, GPU call 22: stem kernel -- correctness, micro-benchmark vs MIOpen, headline bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_igemm_gpu.py -q --tb=short -k "stem" 2>&1 | tail -12 | cut -c1-300
timeout 300 python tools/gpu/stem_bench.py 2>&1 | tail -2 | cut -c1-300
timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c22_bench.log 2>&1; grep '"metric"' gpurun_out/r02_c22_bench.log | cut -c100-260
