# round 2, GPU call 3: integration of the new kernels in the learner step -- tests, headline bench, per-step kernel table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -q --tb=short 2>&1 | tail -12 | cut -c1-300 > gpurun_out/r02_c3_tests.log; cat gpurun_out/r02_c3_tests.log
SHAPES="56,64,64,0;56,64,256,1;56,256,64,0;56,256,128,0;28,128,512,1;14,256,1024,1;14,1024,256,0;7,512,2048,1" timeout 600 python tools/gpu/conv_bench2.py > gpurun_out/r02_c3_conv_bench2.log 2>&1; tail -9 gpurun_out/r02_c3_conv_bench2.log | cut -c1-200
timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r02_c3_bench.log 2>&1; tail -1 gpurun_out/r02_c3_bench.log | cut -c1-900
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_c3_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find /tmp/prof_c3 -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r02_c3_step_kernels.csv | head -40 | cut -c1-170
