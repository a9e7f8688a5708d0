# First gpurun call of the next round (≈ 8 GPU-minutes).  Round 2 ended with one fix that could only be verified on the CPU
# (the reference cycle that kept a step's activations alive until Python's cyclic collector ran), so
# the first thing to establish is that EVERY bench process is now healthy and that the allocator no longer grows:
#   gpurun --timeout 900 -- 'bash tools/gpu/next_round_first_call.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { python -c "
import json,sys
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    d=json.loads(ln); m=d['memory']
    print('$1', round(d['value']), 'img/s', round(d['ms_per_step'],2), 'ms/step | host share', round(d['launch_probe']['after_warmup']['host_share_of_two_steps'],2),
          '| reserved GB', round(m['after_warmup']['reserved_gb'],1), '->', round(m['after_timed_region']['reserved_gb'],1),
          '| segments', m['after_warmup']['segments_allocated'], '->', m['after_timed_region']['segments_allocated'], '| retries', m['after_timed_region']['alloc_retries'])"; }
# 1. five bench processes in a row, the first one on the fresh box (≈ 2.5 min).  Expected: ~8.8-9.0 k img/s each, host share
#    ~0.45, reserved memory and segment count CONSTANT across the timed region (round 2: 26 -> 52 GB, 403 -> 823 segments)
for i in 1 2 3 4 5; do timeout 300 python bench.py --no_cpu_baseline 2>gpurun_out/r03_first_bench_$i.err | show "bench $i:"; done
ls gpurun_out/bench_host_bound_profile.txt 2>/dev/null && head -30 gpurun_out/bench_host_bound_profile.txt
# 2. correctness gate (≈ 4.5 min)
timeout 1800 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -5 | cut -c1-300 > gpurun_out/r03_pytest.log; cat gpurun_out/r03_pytest.log
# 3. kernel trace of the step for the per-step table (≈ 60 s)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r3a -o r3a -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find /tmp/prof_r3a -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r03_step_kernels.csv | head -24 | cut -c1-160
cp $(find /tmp/prof_r3a -name '*kernel_stats.csv' | head -1) gpurun_out/r03_kernel_stats.csv
