# (RECORD of what ran: the switches PF_TEACHER_SHARE / PF_TEACHER_CU_MASK / PF_IGEMM_PP it sets were deleted with the losing code after this call;
#  results: profiles/r06_teacher_share_ab.txt, profiles/r06_bn_bwd_apply_fold_ceiling.txt)
# Round 6, GPU call 2: VERDICT r5 next #3 -- partition the chip between student and teacher, then dispatch or delete the ping-pong kernel.
# One box, every configuration the same command: python bench.py --steps 20 --warmup 5 --no_cpu_baseline  (recorded step; images/s).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
run() {  # label, env...
  label=$1; shift
  v=$(env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); sg = d['config'].get('step_graph'); print('%.0f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], 'recorded' if sg else 'launch by launch'))
")
  echo "$label | $v" | tee -a $O/r06_teacher_share_ab.txt
  [ -z "$v" ] && tail -3 $O/r06_ab_err.txt
}
rm -f $O/r06_teacher_share_ab.txt
echo "# student / teacher partition A/B, one box, bench.py --steps 20 --warmup 5 --no_cpu_baseline (ResNet-50 UQ w8/a8 + dst, B = 256, recorded step unless noted)" >> $O/r06_teacher_share_ab.txt
run "default (teacher at the whole chip, per-tap 3x3)        " PF_X=0
run "teacher share 0.50                                      " PF_TEACHER_SHARE=0.5
run "teacher share 0.33                                      " PF_TEACHER_SHARE=0.33
run "teacher share 0.25                                      " PF_TEACHER_SHARE=0.25
run "ping-pong 3x3 (PF_IGEMM_PP=1)                           " PF_IGEMM_PP=1
run "ping-pong 3x3 + teacher share 0.50                      " PF_IGEMM_PP=1 PF_TEACHER_SHARE=0.5
run "ping-pong 3x3 + teacher share 0.33                      " PF_IGEMM_PP=1 PF_TEACHER_SHARE=0.33
run "ping-pong 3x3 + teacher share 0.25                      " PF_IGEMM_PP=1 PF_TEACHER_SHARE=0.25
run "default again (drift of the box)                        " PF_X=0

runl() {  # launch-by-launch variants (--step_graph 0): the only mode in which a CU-masked stream keeps its mask
  label=$1; shift
  v=$(env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --step_graph 0 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('%.0f images/s  %.2f ms/step  launch by launch (host submit median %.1f ms)' % (d['value'], d['ms_per_step'], d['host_submit_ms_min_median_max'][1]))
")
  echo "$label | $v" | tee -a $O/r06_teacher_share_ab.txt
  [ -z "$v" ] && tail -3 $O/r06_ab_err.txt
}
runl "launch by launch: default                               " PF_X=0
runl "launch by launch: teacher stream on 2 of 8 XCDs         " PF_TEACHER_CU_MASK=xcd:2
runl "launch by launch: teacher stream on 1 of 8 XCDs         " PF_TEACHER_CU_MASK=xcd:1
runl "launch by launch: 2 XCDs + ping-pong 3x3                " PF_TEACHER_CU_MASK=xcd:2 PF_IGEMM_PP=1
runl "launch by launch: teacher share 0.33 + ping-pong 3x3    " PF_TEACHER_SHARE=0.33 PF_IGEMM_PP=1
run "recorded: teacher stream on 2 of 8 XCDs (mask at capture)" PF_TEACHER_CU_MASK=xcd:2
# where the 25 ms of a REPLAYED step go: kernel trace of the default command, the 4 fastest consecutive steps (= replays)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_rec -o rec -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no_cpu_baseline > $GRAFT_REPO_ROOT/$O/r06_prof_rec.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find /tmp/prof_rec -name '*kernel_trace.csv' | head -1) --steps 4 --out $O/r06_step_kernels_recorded_before.csv | head -12 | cut -c1-200
head -12 $O/r06_step_kernels_recorded_before.csv | cut -c1-200
# ceiling of folding k_bn_bwd_apply into its consumers (garbage results by design: tools/gpu/ablate_bn_apply.py)
rm -f $O/r06_bn_bwd_apply_fold_ceiling.txt
for mode in none foldable all none; do
  v=$(timeout 400 python tools/gpu/ablate_bn_apply.py $mode --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('%.0f images/s  %.2f ms/step' % (d['value'], d['ms_per_step']))
")
  echo "apply passes dropped: $mode | $v | $(grep ablate_bn_apply $O/r06_ab_err.txt | cut -c1-160)" | tee -a $O/r06_bn_bwd_apply_fold_ceiling.txt
done
timeout 300 python tools/gpu/bn_bwd_bench.py 2>/dev/null | cut -c1-40,100-200 > $O/r06_bn_bwd_bench.txt; tail -15 $O/r06_bn_bwd_bench.txt
