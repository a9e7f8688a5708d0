# round 2, GPU call 23: is the stem kernel on the step path?  A/B against MIOpen in one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 1 0 1 0; do
  PF_OWN_STEM=$v timeout 600 python bench.py --no_cpu_baseline 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('PF_OWN_STEM=$v', round(d['value']), round(d['ms_per_step'],2))"
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c23 -o c23 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --batch 256 --no_cpu_baseline > /tmp/c23.log 2>&1)
python tools/prof_summary.py $(find /tmp/prof_c23 -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r02_c23_step_kernels.csv | grep -i -E "steady|category|stem|igemm_fwd_gtc|igemm_wrw_gtc|SubTensor|ck::" | cut -c1-170
