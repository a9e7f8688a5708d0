# First gpurun call of the next round (≈ 22 GPU-minutes: builds 2, seven bench runs 8, three test runs 6, layer / ablation tables 6).  Everything is taken in ONE box (the round-3 boxes differed by up to 7 % for
# the same code, so cross-box comparisons mean nothing):
#   gpurun --timeout 2400 -- 'bash tools/gpu/next_round_first_call.sh'
# Round 3 ended with three questions that only the GPU answers; the builds for them are ready:
#   A. ISA: hipcc serialises the LDS fragment reads of the contraction loops -- read, lgkmcnt(0), 4 MFMAs, read, ...
#      (k_igemm), read, lgkmcnt(1), ONE MFMA (k_conv1x1_stream) -- an exposed LDS round trip per group.  The variant library
#      (-DPF_IG_SGB -DPF_ST_SGB -DPF_W2_SGB -DPF_RAW_MINMAX: sched_group_barrier pipelines, batched epilogue reads, bare v_min / v_max in the statistics; same arithmetic in the same order) must be bit-identical; is it faster?
#   B. What does the epilogue cost (ablation builds 4 / 5), and how far is the main loop's matrix work from the tile schedule's ideal?
#   C. Two workgroups per CU run in lockstep (both in their epilogue at once): does starting the second one late help?
#   D. Every persistent launch ends with idle CUs (ceil(tiles / slots) rounds): does a second queue -- the frozen teacher's forward
#      pass over the NEXT batch -- fill them?  (PF_TEACHER_AHEAD=1)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
[ -f tools/gpu/_build/libig_ablate5.so ] || bash tools/gpu/build_ablate.sh > /dev/null 2>&1      # prebuilt libraries travel with the snapshot
[ -f tools/gpu/_build/libpocketflow_hip_sgb.so ] || bash tools/gpu/build_variant.sh sgb -DPF_IG_SGB -DPF_ST_SGB -DPF_W2_SGB -DPF_RAW_MINMAX > /dev/null 2>&1
V=$GRAFT_REPO_ROOT/tools/gpu/_build/libpocketflow_hip_sgb.so
# the same + (three-stage prologue kernels) the residual vectors of a conv3 tile requested one k-step earlier (-DPF_IG_RES_EARLY) and
# both halves' fragments read in front of the first half's MFMAs (-DPF_IG_SGB_PRO2: 256 registers, no spill); the 128-wide resident kernel with two input chunks in flight instead of three (-DPF_ST_D128=2: no spills; 3-21 at depth 3)
[ -f tools/gpu/_build/libpocketflow_hip_sgb2.so ] || bash tools/gpu/build_variant.sh sgb2 -DPF_IG_SGB -DPF_ST_SGB -DPF_W2_SGB -DPF_RAW_MINMAX -DPF_IG_RES_EARLY -DPF_IG_SGB_PRO2 -DPF_ST_D128=2 > /dev/null 2>&1
V2=$GRAFT_REPO_ROOT/tools/gpu/_build/libpocketflow_hip_sgb2.so
line() { python -c "
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d = json.loads(ln); print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms/step | host submit', d.get('host_submit_ms_min_median_max'), '| roofline frac', d['roofline']['frac'], '| region ms per launch', d['roofline'].get('avg_launch_ms'))
" $1 $2; }
# 1. the step: product library, variant library, product again (drift check)
for tag in product sgb sgb2 product2; do
  lib=""; [ $tag = sgb ] && lib=$V; [ $tag = sgb2 ] && lib=$V2
  PF_HIP_LIB=$lib timeout 400 python bench.py --steps 15 --warmup 5 --no_cpu_baseline > gpurun_out/r04_first_bench_$tag.json 2> gpurun_out/r04_first_bench_$tag.err || tail -3 gpurun_out/r04_first_bench_$tag.err
  line gpurun_out/r04_first_bench_$tag.json $tag
done
# 1b. D: the teacher's forward over batch k+1 on a second stream beside step k's backward (learners/teacher_ahead.py, opt-in): the
#     step with it, with it AND the variant library, and the distillation parity tests with it (a stream race would show there)
PF_TEACHER_AHEAD=1 timeout 400 python bench.py --steps 15 --warmup 5 --no_cpu_baseline > gpurun_out/r04_first_bench_ahead.json 2> gpurun_out/r04_first_bench_ahead.err || tail -3 gpurun_out/r04_first_bench_ahead.err
line gpurun_out/r04_first_bench_ahead.json teacher-ahead
PF_TEACHER_AHEAD=1 PF_HIP_LIB=$V timeout 400 python bench.py --steps 15 --warmup 5 --no_cpu_baseline > gpurun_out/r04_first_bench_ahead_sgb.json 2> gpurun_out/r04_first_bench_ahead_sgb.err || tail -3 gpurun_out/r04_first_bench_ahead_sgb.err
line gpurun_out/r04_first_bench_ahead_sgb.json teacher-ahead+sgb
PF_TEACHER_AHEAD=1 timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x --tb=line -k "distillation or bf16_fused or conditioned" 2>&1 | tail -4 | cut -c1-300 | tee gpurun_out/r04_first_pytest_ahead.log
PF_TEACHER_AHEAD=1 timeout 400 python bench.py --config c4 --steps 10 --warmup 4 --no_cpu_baseline > gpurun_out/r04_first_bench_c4_ahead.json 2> gpurun_out/r04_first_bench_c4_ahead.err; line gpurun_out/r04_first_bench_c4_ahead.json c4-teacher-ahead
timeout 400 python bench.py --config c4 --steps 10 --warmup 4 --no_cpu_baseline > gpurun_out/r04_first_bench_c4.json 2> gpurun_out/r04_first_bench_c4.err; line gpurun_out/r04_first_bench_c4.json c4
# 2. A: per layer, bit-identity asserted; then the convolution test files against the variant library
timeout 400 python tools/gpu/igemm_sgb_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_first_igemm_sgb.txt | cut -c1-200
PF_HIP_LIB=$V timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -m gpu -q -x --tb=line 2>&1 | tail -4 | cut -c1-300 | tee gpurun_out/r04_first_pytest_sgb.log
PF_HIP_LIB=$V timeout 400 python tools/gpu/fwd1x1_layers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_first_fwd1x1_layers_sgb.txt
PF_HIP_LIB=$V2 timeout 400 python tools/gpu/fwd1x1_layers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_first_fwd1x1_layers_sgb2.txt
PF_HIP_LIB=$V2 timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -m gpu -q -x --tb=line 2>&1 | tail -3 | cut -c1-300 | tee gpurun_out/r04_first_pytest_sgb2.log
timeout 400 python tools/gpu/fwd1x1_layers.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_first_fwd1x1_layers.txt | cut -c1-200
# 3. B: ablations incl. the no-epilogue builds and the tile schedule's ideal
TILES=128x128 timeout 400 python tools/gpu/igemm_ablate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_first_igemm_ablation.txt | cut -c1-220
# 4. C: second workgroup of every CU started late
timeout 400 python tools/gpu/igemm_stagger.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_first_igemm_stagger.txt | cut -c1-200
# 5. the host-bound CIFAR-size configuration and backward-filter per layer (baselines)
timeout 400 python bench.py --config c1 --steps 15 --warmup 5 --no_cpu_baseline > gpurun_out/r04_first_bench_c1.json 2> gpurun_out/r04_first_bench_c1.err; line gpurun_out/r04_first_bench_c1.json c1
TARGETS=256,384 timeout 300 python tools/gpu/wrw_target_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_first_wrw.txt | cut -c1-200
# afterwards, here:  python tools/first_call_report.py gpurun_out r04_first
