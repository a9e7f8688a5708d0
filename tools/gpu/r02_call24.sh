# round 2, GPU call 24: stem backward-filter -- correctness, micro-benchmark vs MIOpen, A/B on the step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_igemm_gpu.py -q --tb=short -k "stem" 2>&1 | tail -12 | cut -c1-300
timeout 300 python tools/gpu/stem_bench.py 2>&1 | tail -2 | cut -c1-300
for v in 1 0; do
  PF_OWN_CONV2D_WRW=$v timeout 600 python bench.py --no_cpu_baseline 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('PF_OWN_CONV2D_WRW=$v', round(d['value']), round(d['ms_per_step'],2))"
done
