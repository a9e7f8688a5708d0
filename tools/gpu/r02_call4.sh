# round 2, GPU call 4: new backward-filter kernel, max-pool kernels, updated dispatch -- tests, A/B, headline bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py tests/test_kernels_gpu.py -q --tb=short -k "wrw or maxpool or prologue_fake or strided" 2>&1 | tail -20 | cut -c1-300 > gpurun_out/r02_c4_tests.log; cat gpurun_out/r02_c4_tests.log
timeout 600 python tools/gpu/wrw_bench.py > gpurun_out/r02_c4_wrw_bench.log 2>&1; tail -24 gpurun_out/r02_c4_wrw_bench.log | cut -c1-200
timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c4_bench.log 2>&1; tail -1 gpurun_out/r02_c4_bench.log | cut -c1-600
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_c4_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find /tmp/prof_c4 -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r02_c4_step_kernels.csv | head -36 | cut -c1-150
timeout 900 python -m pytest tests/test_learner_gpu.py tests/test_conv_gpu.py -q --tb=short -k "as_accurate or resnet or uq" 2>&1 | tail -8 | cut -c1-400 > gpurun_out/r02_c4_learner.log; cat gpurun_out/r02_c4_learner.log
