# Round 6, GPU call 16: backward-filter launches on a second queue (graph.WrwSide) -- equality tests, step A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests/test_learner_gpu.py -m gpu -q --tb=short -x -k "side_queue or step_graph_is_the_eager" 2>&1 | tail -8 | cut -c1-400
run() {  # label, extra bench flags ("-" for none), env...
  label=$1; flags=$2; shift; shift
  [ "$flags" = "-" ] && flags=""
  v=$(env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline $flags 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); sg = d['config'].get('step_graph'); print('%.0f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], 'recorded' if sg else 'launch by launch'))
")
  echo "$label | $v" | tee -a $O/r06_wrw_side_ab.txt
  [ -z "$v" ] && tail -5 $O/r06_ab_err.txt
}
rm -f $O/r06_wrw_side_ab.txt
run "one queue (PF_WRW_SIDE=0)            " - PF_WRW_SIDE=0
run "filter launches on a second queue    " - PF_X=0
run "one queue (PF_WRW_SIDE=0)            " - PF_WRW_SIDE=0
run "filter launches on a second queue    " - PF_X=0
run "one queue, launch by launch          " "--step_graph 0" PF_WRW_SIDE=0
run "second queue, launch by launch       " "--step_graph 0" PF_X=0
exit 0
