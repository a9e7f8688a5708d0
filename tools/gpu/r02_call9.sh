# round 2, GPU call 9: wide prologue tiles A/B; headline bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_gpu.py -q --tb=short -k "prologue or plain_and_residual" 2>&1 | tail -4 | cut -c1-300
for w in 1 0; do echo "PF_IGEMM_PRO_WIDE=$w"; PF_IGEMM_PRO_WIDE=$w SHAPES="14,256,1024,1;28,256,512,0;14,512,1024,0" timeout 600 python tools/gpu/conv_bench2.py 2>&1 | tail -3 | cut -c1-120; done
timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c9_bench.log 2>&1; tail -1 gpurun_out/r02_c9_bench.log | cut -c1-300
PF_IGEMM_PRO_WIDE=0 timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c9_bench_nowide.log 2>&1; tail -1 gpurun_out/r02_c9_bench_nowide.log | cut -c1-300
