# Round 6, GPU call 3: regression of the kernel tests after the deletions, then dispatch-policy A/Bs in one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1200 python -m pytest tests/test_igemm_gpu.py tests/test_conv_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q --tb=short 2>&1 | tail -6 | cut -c1-300
run() {  # label, env...
  label=$1; shift
  v=$(env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); sg = d['config'].get('step_graph'); print('%.0f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], 'recorded' if sg else 'launch by launch'))
")
  echo "$label | $v" | tee -a $O/r06_policy_ab.txt
  [ -z "$v" ] && tail -3 $O/r06_ab_err.txt
}
rm -f $O/r06_policy_ab.txt
echo "# dispatch-policy A/B, one box, bench.py --steps 20 --warmup 5 --no_cpu_baseline (ResNet-50 UQ w8/a8 + dst, B = 256)" >> $O/r06_policy_ab.txt
run "default                                                  " PF_X=0
run "bn3 materialised from C = 512 (stage 4)                  " PF_BN3_MATERIALIZE_MIN_C=512
run "bn3 materialised from C = 256 (stages 3-4)               " PF_BN3_MATERIALIZE_MIN_C=256
run "bn3 materialised from C = 128 (stages 2-4)               " PF_BN3_MATERIALIZE_MIN_C=128
run "bn3 materialised everywhere                              " PF_BN3_MATERIALIZE_MIN_C=1
run "teacher branch forked at the start of the backward pass  " PF_TEACHER_FORK=backward
run "bn3 from C = 256 + teacher forked at backward            " PF_BN3_MATERIALIZE_MIN_C=256 PF_TEACHER_FORK=backward
run "default again (drift of the box)                         " PF_X=0
# parity of the materialised-bn3 path at step level (the conditioned ResNet-50 tests, bf16 fused path)
PF_BN3_MATERIALIZE_MIN_C=256 timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -k "bf16_fused_path or one_step_at_224" 2>&1 | tail -4 | cut -c1-300
rm -f $O/r06_bn_bwd_apply_fold_ceiling2.txt
for mode in none foldable none; do
  v=$(timeout 400 python tools/gpu/ablate_bn_apply.py $mode --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('%.0f images/s  %.2f ms/step' % (d['value'], d['ms_per_step']))
")
  echo "apply passes dropped: $mode | $v | $(grep ablate_bn_apply $O/r06_ab_err.txt | cut -c1-160)" | tee -a $O/r06_bn_bwd_apply_fold_ceiling2.txt
done
timeout 300 python tools/gpu/bn_bwd_bench.py 2>/dev/null | cut -c1-34,100-200 > $O/r06_bn_bwd_bench.txt; tail -15 $O/r06_bn_bwd_bench.txt
