cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_learner_gpu.py -m gpu -q --timeout=300 --tb=short -k "fused_path or channel_pruned" 2>&1 | tail -30 > gpurun_out/pytest9.log
cat gpurun_out/pytest9.log
NO_MIOPEN=1 VARIANTS=1 timeout 300 python tools/gpu/conv_bench.py 2>&1 | grep -A1 "^56,64,256\|^56,256,64\|^28,512,128\|^14,1024" 
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r1d -o r1d -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1d.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/prof_r1d.log | cut -c1-300
python tools/prof_summary.py $(find /tmp/prof_r1d -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r01_step_kernels_fused.csv | head -45 | cut -c1-180
