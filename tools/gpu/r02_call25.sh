# round 2, GPU call 25: bench with the library warm-up child (first process on the box); two-consumer join A/B; 2-rank bench test
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', round(d['value']), round(d['ms_per_step'],2), 'host', [round(v,1) for v in d['host_submit_ms_min_median_max']], d['launch_probe'])"; }
t0=$(date +%s)
timeout 900 python bench.py --no_cpu_baseline 2>/dev/null | grep '"metric"' | show "default(prewarm)"
echo "wall of the default bench: $(( $(date +%s) - t0 )) s"
for v in 1 0 1 0; do
  PF_JOIN_TWO_CONSUMERS=$v timeout 600 python bench.py --no_cpu_baseline --no_prewarm 2>/dev/null | grep '"metric"' | show "PF_JOIN_TWO_CONSUMERS=$v"
done
timeout 900 python -m pytest tests/test_learner_gpu.py -q --tb=short -k "two_ranks" 2>&1 | tail -3 | cut -c1-300
