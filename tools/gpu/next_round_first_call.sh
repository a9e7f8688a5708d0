# First gpurun call of the next round (≈ 4 GPU-minutes): the baselines every kernel change of round 4 will be measured against,
# taken in ONE box (the round-3 boxes differed by up to 7 % for the same code, so cross-box comparisons mean nothing):
#   gpurun --timeout 900 -- 'bash tools/gpu/next_round_first_call.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
# 1. the step (default configuration) and the host-bound CIFAR-size configuration
for c in c2 c1; do
  timeout 400 python bench.py --config $c --steps 15 --warmup 5 --no_cpu_baseline > gpurun_out/r04_first_bench_$c.json 2> gpurun_out/r04_first_bench_$c.err || tail -3 gpurun_out/r04_first_bench_$c.err
  python -c "
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d = json.loads(ln); print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms/step | host submit', d.get('host_submit_ms_min_median_max'), '| roofline frac', d['roofline']['frac'])
" gpurun_out/r04_first_bench_$c.json $c
done
# 2. the roofline region layer by layer and one tile of its slowest kernel phase by phase (the k-steps wait on the
#    LDS-DMA fill rate) -- what the next kernel has to beat
timeout 400 python tools/gpu/fwd1x1_layers.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_first_fwd1x1_layers.txt | cut -c1-200
bash tools/gpu/build_ablate.sh > /dev/null 2>&1
timeout 300 python tools/gpu/igemm_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_first_igemm_timeline.txt | cut -c1-400
# 3. backward-filter per layer (kernel + reduction)
TARGETS=256,384 timeout 300 python tools/gpu/wrw_target_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_first_wrw.txt | cut -c1-200
