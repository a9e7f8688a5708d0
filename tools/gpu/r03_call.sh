# gpurun payloads of round 3, by name:  gpurun --timeout N -- 'bash tools/gpu/r03_call.sh <name>'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
# bench.py JSON line (file $2) -> one summary line tagged $1
bench_line() { python -c "
import json, sys
for ln in open(sys.argv[2]):
    if not ln.startswith('{'): continue
    d = json.loads(ln)
    print(sys.argv[1], round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms/step | host submit', d.get('host_submit_ms_min_median_max'), '| roofline frac', d.get('roofline', {}).get('frac'))
" "$1" "$2"; }
run_bench() {  # $1 tag, rest: bench.py arguments
  tag=$1; shift
  timeout 400 python bench.py "$@" > gpurun_out/r03_bench_$tag.json 2> gpurun_out/r03_bench_$tag.err || tail -5 gpurun_out/r03_bench_$tag.err
  bench_line $tag gpurun_out/r03_bench_$tag.json
}
case "$1" in
c1)
  # new prologue arithmetic + 3-stage prologue kernel: numerics, per-layer A/B, ablation of the plain kernel, step A/B
  timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -25 | cut -c1-400 > gpurun_out/r03_c1_pytest.log
  tail -8 gpurun_out/r03_c1_pytest.log
  timeout 300 python tools/gpu/pro_bench.py > gpurun_out/r03_c1_pro_bench.log 2>&1; cat gpurun_out/r03_c1_pro_bench.log
  timeout 400 python tools/gpu/igemm_ablate.py > gpurun_out/r03_c1_ablate.log 2>&1; cat gpurun_out/r03_c1_ablate.log
  PF_IGEMM_PRO3=1 run_bench c1_pro3 --steps 10 --warmup 5 --no_cpu_baseline
  PF_IGEMM_PRO3=0 run_bench c1_pro2 --steps 10 --warmup 5 --no_cpu_baseline
  ;;
c2)
  # full GPU suite (new gradient-level parity tests; report lines -> gpurun_out/r03_parity_report.txt), the headline bench with
  # the 3-stage prologue kernel on and off, and the other concrete runs of SURVEY 8(d)
  rm -f gpurun_out/r03_parity_report.txt
  PF_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r03_parity_report.txt timeout 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 | cut -c1-600 > gpurun_out/r03_c2_pytest.log
  tail -12 gpurun_out/r03_c2_pytest.log; cat gpurun_out/r03_parity_report.txt
  PF_IGEMM_PRO3=1 run_bench c2_pro3 --steps 15 --warmup 5 --no_cpu_baseline
  PF_IGEMM_PRO3=0 run_bench c2_pro2 --steps 15 --warmup 5 --no_cpu_baseline
  for c in c2a32 c4 c3 c1; do run_bench cfg_$c --config $c --steps 10 --warmup 4 --no_cpu_baseline; done
  ;;
c3)
  # the step-level parity tests (gradient level) + the 2-rank bench test, then the benches of all configurations
  rm -f gpurun_out/r03_parity_report.txt
  PF_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r03_parity_report.txt timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_learner_gpu.py -m gpu -q --tb=short 2>&1 | tail -30 | cut -c1-600 > gpurun_out/r03_c3_pytest.log
  tail -8 gpurun_out/r03_c3_pytest.log; cat gpurun_out/r03_parity_report.txt
  PF_IGEMM_PRO3=1 run_bench c2_pro3 --steps 15 --warmup 5 --no_cpu_baseline
  PF_IGEMM_PRO3=0 run_bench c2_pro2 --steps 15 --warmup 5 --no_cpu_baseline
  for c in c2a32 c4 c3 c1; do run_bench cfg_$c --config $c --steps 10 --warmup 4 --no_cpu_baseline; done
  ;;
c4)
  # depthwise kernels + the 2-rank control flow, backward-filter ablation, per-step kernel table, MobileNet bench with own depthwise
  timeout 900 python -m pytest tests/test_depthwise_gpu.py tests/test_learner_gpu.py -m gpu -q --tb=short -k "depthwise or two_ranks or mobilenet" 2>&1 | tail -15 | cut -c1-500 > gpurun_out/r03_c4_pytest.log
  tail -6 gpurun_out/r03_c4_pytest.log
  timeout 400 python tools/gpu/wrw_ablate.py > gpurun_out/r03_wrw_ablation.txt 2>&1; cat gpurun_out/r03_wrw_ablation.txt
  run_bench cfg_c3 --config c3 --steps 10 --warmup 4 --no_cpu_baseline
  PF_OWN_DEPTHWISE=0 run_bench cfg_c3_miopen --config c3 --steps 10 --warmup 4 --no_cpu_baseline
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r3a -o r3a -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_prof.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/prof_summary.py $(find /tmp/prof_r3a -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r03_step_kernels_b256.csv | head -40 | cut -c1-170
  cp $(find /tmp/prof_r3a -name '*kernel_stats.csv' | head -1) gpurun_out/r03_rocprofv3_stats_b256.csv
  ;;
c5)
  # wave-specialised prologue kernel: numerics, per-layer A/B (graph-replay timing), ablations re-timed without host overhead, step A/B
  timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -x -k "three_stage or prologue" 2>&1 | tail -12 | cut -c1-400 > gpurun_out/r03_c5_pytest.log
  tail -5 gpurun_out/r03_c5_pytest.log
  timeout 300 python tools/gpu/pro_bench.py > gpurun_out/r03_pro_bench.txt 2>&1; cat gpurun_out/r03_pro_bench.txt
  timeout 400 python tools/gpu/wrw_ablate.py > gpurun_out/r03_wrw_ablation.txt 2>&1; cat gpurun_out/r03_wrw_ablation.txt
  timeout 400 python tools/gpu/igemm_ablate.py > gpurun_out/r03_igemm_ablation.txt 2>&1; cat gpurun_out/r03_igemm_ablation.txt
  PF_IGEMM_PROW=1 run_bench c2_ws --steps 15 --warmup 5 --no_cpu_baseline
  PF_IGEMM_PROW=0 run_bench c2_pro3 --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c6)
  timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -k "three_stage" 2>&1 | tail -6 | cut -c1-300
  timeout 300 python tools/gpu/pro_bench.py > gpurun_out/r03_pro_bench.txt 2>&1; cat gpurun_out/r03_pro_bench.txt
  timeout 400 python tools/gpu/wrw_ablate.py > gpurun_out/r03_wrw_ablation.txt 2>&1; cat gpurun_out/r03_wrw_ablation.txt
  timeout 400 python tools/gpu/igemm_ablate.py > gpurun_out/r03_igemm_ablation.txt 2>&1; cat gpurun_out/r03_igemm_ablation.txt
  ;;
c7)
  # three-stage backward-filter kernel (asm transposing reads, counted vmcnt): numerics, ablation, step
  timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -m gpu -q --tb=short -k "wrw or backward_filter" 2>&1 | tail -6 | cut -c1-300
  timeout 400 python tools/gpu/wrw_ablate.py > gpurun_out/r03_wrw_ablation.txt 2>&1; cat gpurun_out/r03_wrw_ablation.txt
  run_bench c2_wrw3 --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c8)
  timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -m gpu -q --tb=short -k "wrw or backward_filter" 2>&1 | tail -4 | cut -c1-300
  timeout 300 python tools/gpu/wrw_timeline.py 2>&1 | tee gpurun_out/r03_wrw_timeline.txt
  ;;
c12)
  timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -m gpu -q --tb=short -k "wrw or backward_filter" 2>&1 | tail -4 | cut -c1-300
  timeout 300 python tools/gpu/wrw_timeline.py 2>&1 | tee gpurun_out/r03_wrw_timeline.txt
  timeout 400 python tools/gpu/wrw_ablate.py 2>&1 | tee gpurun_out/r03_wrw_ablation.txt
  run_bench c2_wrw3b --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c13)
  # halo 3x3 kernel: numerics (forward, statistics + residual, backward-data + BN sums), per-layer table vs per-tap igemm / MIOpen, step
  timeout 900 python -m pytest tests/test_igemm_gpu.py -m gpu -q --tb=short -x -k "conv2d_fwd or backward_data" 2>&1 | tail -12 | cut -c1-300
  timeout 600 python tools/gpu/igemm_bench.py 2>&1 | tee gpurun_out/r03_igemm_layers.txt | cut -c1-220
  PF_CONV3X3_HALO=1 run_bench c2_halo --steps 15 --warmup 5 --no_cpu_baseline
  PF_CONV3X3_HALO=0 run_bench c2_nohalo --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c14)
  # residual on the fp32 accumulators (all four forward kernels): numerics, per-layer cost, the roofline region layer by layer, step
  timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -12 | cut -c1-300
  timeout 300 python tools/gpu/pro_bench.py 2>&1 | tee gpurun_out/r03_pro_bench_res32.txt | cut -c1-220
  timeout 400 python tools/gpu/fwd1x1_layers.py 2>&1 | tee gpurun_out/r03_fwd1x1_layers.txt | cut -c1-220
  run_bench c2_res32 --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c15)
  # residual prefetch in the tiled kernel + resident-kernel variant with 4-32 column slices: numerics, the roofline region layer by
  # layer under both plans, step A/B, and the idle-gap analysis of the step
  timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -12 | cut -c1-300
  PF_CONV_STREAM_MAXSPLIT=2 timeout 400 python tools/gpu/fwd1x1_layers.py 2>&1 | tee gpurun_out/r03_fwd1x1_layers_split2.txt | cut -c1-220
  PF_CONV_STREAM_MAXSPLIT=32 timeout 400 python tools/gpu/fwd1x1_layers.py 2>&1 | tee gpurun_out/r03_fwd1x1_layers_split32.txt | cut -c1-220
  PF_CONV_STREAM_MAXSPLIT=2 run_bench c2_split2 --steps 15 --warmup 5 --no_cpu_baseline
  PF_CONV_STREAM_MAXSPLIT=32 run_bench c2_split32 --steps 15 --warmup 5 --no_cpu_baseline
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r3b -o r3b -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_prof_b.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/prof_summary.py $(find /tmp/prof_r3b -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r03_step_kernels_b256_c15.csv | head -16 | cut -c1-200
  ;;
c16)
  # full GPU suite on the current tree (a17 GPU tests, residual changes, many-slice stream test), host-side API view of the
  # steady state (what the host does during the device's idle gaps), own 3x3 backward-filter at C = 64 vs MIOpen in the step
  rm -f gpurun_out/r03_parity_report.txt
  PF_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r03_parity_report.txt timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --tb=short 2>&1 | tail -30 | cut -c1-400 > gpurun_out/r03_c16_pytest.log
  tail -14 gpurun_out/r03_c16_pytest.log
  cd /tmp
  timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/prof_r3h -o r3h -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_prof_h.log 2>&1
  cd $GRAFT_REPO_ROOT
  ls /tmp/prof_r3h/*/ 2>/dev/null | head; head -2 $(find /tmp/prof_r3h -name '*hip_api_trace.csv' | head -1) | cut -c1-300
  python tools/hip_api_summary.py $(find /tmp/prof_r3h -name '*hip_api_trace.csv' | head -1) $(find /tmp/prof_r3h -name '*kernel_trace.csv' | head -1) --steps 4 2>&1 | tee gpurun_out/r03_hip_api_summary.txt | cut -c1-400
  PF_OWN_CONV2D_WRW_MIN_C=64 run_bench c2_wrw64own --steps 15 --warmup 5 --no_cpu_baseline
  run_bench c2_wrw64miopen --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c17)
  # the two parity tests that failed in the full run: alone, then behind the a17 tests (flag leakage?); new max-pool kernels
  timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=line -k "conditioned_state or cp_mobilenet" 2>&1 | tail -8 | cut -c1-300
  timeout 900 python -m pytest tests/test_learner_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=line -k "cp_feature_sampling or conditioned_state or cp_mobilenet" 2>&1 | tail -8 | cut -c1-300
  PF_OWN_DEPTHWISE=0 timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=line -k "cp_mobilenet" 2>&1 | tail -5 | cut -c1-300
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "maxpool" 2>&1 | tail -6 | cut -c1-300
  PF_POOL3S2=1 run_bench c2_pool1 --steps 15 --warmup 5 --no_cpu_baseline
  PF_POOL3S2=0 run_bench c2_pool0 --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c18)
  # BN-backward apply: rows per trip x grid sizing
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "bn" 2>&1 | tail -4 | cut -c1-300
  timeout 400 python tools/gpu/bn_bwd_bench.py 2>&1 | tee gpurun_out/r03_bn_bwd_bench.txt | cut -c1-200
  ;;
c19)
  # final-state GPU suite (integer-export tests, hermetic conditioned-state test) + one default bench line with the committed PMC traffic
  rm -f gpurun_out/r03_parity_report.txt
  PF_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r03_parity_report.txt timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --tb=short 2>&1 | tail -30 | cut -c1-400 > gpurun_out/r03_pytest_gpu.log
  tail -8 gpurun_out/r03_pytest_gpu.log
  run_bench final --steps 20 --warmup 5 --no_cpu_baseline
  python -c "
import json
d = json.loads([l for l in open('gpurun_out/r03_bench_final.json') if l.startswith('{')][0]); print(d['roofline'])"
  ;;
c20)
  # backward-filter: workgroups per launch (pixel splits -> slab traffic vs parallelism), kernel + reduction per layer
  timeout 600 python tools/gpu/wrw_target_bench.py 2>&1 | tee gpurun_out/r03_wrw_target_bench.txt | cut -c1-200
  ;;
c21)
  # backward-filter split targets below one workgroup per CU, numerics with the new defaults, step A/B
  TARGETS=128,192,256,320,384 timeout 600 python tools/gpu/wrw_target_bench.py 2>&1 | tee gpurun_out/r03_wrw_target_bench_low.txt | cut -c1-200
  timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -m gpu -q --tb=short -k "wrw" 2>&1 | tail -4 | cut -c1-300
  run_bench c2_wrwtarget_new --steps 15 --warmup 5 --no_cpu_baseline
  PF_WRW2_TARGET=512 run_bench c2_wrwtarget_512 --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c22)
  # igemm: 1x1 fast path of the tile setup + cross-tile prefetch of the three-stage kernels: numerics, race screen, per-layer, step
  timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -8 | cut -c1-300
  timeout 400 python tools/gpu/fwd1x1_layers.py 2>&1 | tee gpurun_out/r03_fwd1x1_layers_xpre.txt | cut -c1-220
  run_bench c2_xpre --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c23)
  # phase stagger of the persistent workgroups of k_igemm: the roofline region layer by layer
  for st in 0 3000 6000 12000; do
    echo "== PF_IGEMM_STAGGER=$st"
    PF_IGEMM_STAGGER=$st timeout 300 python tools/gpu/fwd1x1_layers.py 2>&1 | grep -v amdgpu.ids | awk 'NR==1 || /s2 conv1 |s3 conv|s4 conv|proj|per step/' | cut -c1-150
  done | tee gpurun_out/r03_stagger.txt
  ;;
c24)
  # where a tile of the three-stage prologue kernel spends its time
  timeout 300 python tools/gpu/igemm_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_igemm_timeline.txt | cut -c1-400
  ;;
c25)
  # three-stage prologue kernel: the input part of a stage is waited for a step earlier than its kernel part -- numerics incl. the
  # race screen, timeline, the region layer by layer, step
  timeout 1500 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -x -k "three_stage or prologue or stream" 2>&1 | tail -5 | cut -c1-300
  timeout 300 python tools/gpu/igemm_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_igemm_timeline_split.txt | cut -c1-400
  timeout 400 python tools/gpu/fwd1x1_layers.py 2>&1 | tee gpurun_out/r03_fwd1x1_layers_split.txt | cut -c1-220
  run_bench c2_splitwait --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c26)
  # prologue arithmetic on packed FMAs + uniform quant branch: numerics (incl. tie enumeration, wrw prologue), region per layer, step
  timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -5 | cut -c1-300
  timeout 400 python tools/gpu/fwd1x1_layers.py 2>&1 | tee gpurun_out/r03_fwd1x1_layers_pk.txt | cut -c1-220
  run_bench c2_pk --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c27)
  # host side of the host-bound configurations
  CFG=c1 timeout 400 python tools/gpu/host_overhead.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_host_overhead_c1.txt | cut -c1-170
  ;;
c28)
  # raw stream getter: semantics, a kernel subset, the host-bound configurations and the default one
  timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_depthwise_gpu.py -m gpu -q --tb=short 2>&1 | tail -4 | cut -c1-300
  run_bench c1_rawstream --config c1 --steps 20 --warmup 5 --no_cpu_baseline
  run_bench c3_rawstream --config c3 --steps 10 --warmup 4 --no_cpu_baseline
  run_bench c2_rawstream --steps 15 --warmup 5 --no_cpu_baseline
  ;;
c29)
  CFG=c3 timeout 400 python tools/gpu/host_overhead.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_host_overhead_c3.txt | cut -c1-170
  ;;
c30)
  # MobileNet dropout mask through pinned memory (no host/GPU synchronisation per step): parity of the MobileNet steps, C3 bench
  timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_learner_gpu.py -m gpu -q --tb=short -k "mobilenet" 2>&1 | tail -4 | cut -c1-300
  run_bench c3_pinned --config c3 --steps 10 --warmup 4 --no_cpu_baseline
  CFG=c3 timeout 300 python tools/gpu/host_overhead.py 2>&1 | grep -v amdgpu.ids | sed -n 7,14p | cut -c1-170
  ;;
c31)
  timeout 900 python -m pytest tests/test_learner_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=short -k "mobilenet" 2>&1 | tail -25 | cut -c1-300
  run_bench c3_ring --config c3 --steps 10 --warmup 4 --no_cpu_baseline
  CFG=c3 timeout 300 python tools/gpu/host_overhead.py 2>&1 | grep -v amdgpu.ids | sed -n 7,14p | cut -c1-170
  ;;
c32)
  timeout 900 python -m pytest tests/test_learner_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=short -k "mobilenet" 2>&1 | tail -6 | cut -c1-300
  run_bench c3_chunk --config c3 --steps 40 --warmup 4 --no_cpu_baseline
  ;;
c33)
  # learner-level GPU tests on the last commit (raw stream getter, chunked dropout masks)
  rm -f gpurun_out/r03_parity_report_last.txt
  PF_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r03_parity_report_last.txt timeout 1200 python -m pytest tests/test_learner_gpu.py tests/test_parity_gpu.py -m gpu -q --tb=short 2>&1 | tail -6 | cut -c1-300 | tee gpurun_out/r03_pytest_gpu_learners_last.log
  ;;
*) echo "unknown payload $1"; exit 2;;
esac
