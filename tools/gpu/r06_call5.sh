# Round 6, GPU call 5: strided backward-data with fused BN-backward sums (tests + A/B), student stream priority experiment
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests/test_igemm_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q --tb=short -k "strided or conv3x3" 2>&1 | tail -8 | cut -c1-300
run() {  # label, bench args, env...
  label=$1; shift; bargs=$1; shift
  v=$(env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline $bargs 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); sg = d['config'].get('step_graph'); print('%.0f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], 'recorded' if sg else 'launch by launch'))
")
  echo "$label | $v | $(grep 'main stream priority' $O/r06_ab_err.txt | cut -c1-100)" | tee -a $O/r06_call5_ab.txt
  [ -z "$v" ] && tail -3 $O/r06_ab_err.txt
}
rm -f $O/r06_call5_ab.txt
echo "# one box, bench.py --steps 20 --warmup 5 --no_cpu_baseline" >> $O/r06_call5_ab.txt
run "strided bwd-data: separate pf_bn_bwd_stats pass (PF_FUSE_BN_BWD_STATS_STRIDED=0)" "" PF_FUSE_BN_BWD_STATS_STRIDED=0
run "strided bwd-data: sums fused (default)                                        " "" PF_X=0
run "strided bwd-data: separate pass                                               " "" PF_FUSE_BN_BWD_STATS_STRIDED=0
run "strided bwd-data: sums fused (default)                                        " "" PF_X=0
run "recorded, student branch captured on a HIGH-priority stream                   " "" PF_MAIN_STREAM_PRIORITY=high
run "launch by launch, default priorities                                          " "--step_graph 0" PF_X=0
run "launch by launch, student stream HIGH priority                                " "--step_graph 0" PF_MAIN_STREAM_PRIORITY=high
run "launch by launch, default priorities                                          " "--step_graph 0" PF_X=0
run "launch by launch, student stream HIGH priority                                " "--step_graph 0" PF_MAIN_STREAM_PRIORITY=high
