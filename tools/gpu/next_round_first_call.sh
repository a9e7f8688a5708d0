# First gpurun call of the next round (≈ 6 GPU-minutes): re-establish the baseline after this round's additions and
# collect what the kernel work of DESIGN.md section 9 needs, in the order the numbers are needed.
#   gpurun --timeout 600 -- 'bash tools/gpu/next_round_first_call.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
# 1. correctness gate (≈ 2.6 min): everything incl. the searches, the input kernel and chn-pruned-gpu
timeout 900 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -5 | cut -c1-300 > gpurun_out/r02_pytest.log; cat gpurun_out/r02_pytest.log
# 2. headline number (≈ 50 s)
timeout 600 python bench.py > gpurun_out/r02_bench.log 2>&1; tail -1 gpurun_out/r02_bench.log | cut -c1-1200
# 3. where the 1x1 kernels stand per layer, incl. the prologue / residual / statistics increments (≈ 40 s)
VARIANTS=1 timeout 300 python tools/gpu/conv_bench.py > gpurun_out/r02_conv_bench.log 2>&1; tail -30 gpurun_out/r02_conv_bench.log | cut -c1-200
# 4. kernel trace of the step for the per-step table (≈ 60 s): MIOpen's 3x3 kernels are the second-largest block
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2a -o r2a -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find /tmp/prof_r2a -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r02_step_kernels.csv | head -24 | cut -c1-160
cp $(find /tmp/prof_r2a -name '*kernel_stats.csv' | head -1) gpurun_out/r02_kernel_stats.csv
