# Round 6, GPU call 1: parity at the benchmarked geometry first, then the round's "before" tables.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( free -g | head -2; nproc; lscpu | grep 'Model name' ) > $O/r06_box.txt 2>&1
rm -f $O/r06_parity_report_call1.txt
PF_PARITY_REPORT=$GRAFT_REPO_ROOT/$O/r06_parity_report_call1.txt timeout 1500 python -m pytest tests/test_bench_geometry_gpu.py -m gpu -q --timeout=600 --tb=short 2>&1 | tail -40 | cut -c1-400 > $O/r06_bench_geometry.log
tail -5 $O/r06_bench_geometry.log
PF_PARITY_REPORT=$GRAFT_REPO_ROOT/$O/r06_parity_report_call1.txt timeout 2400 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout=1800 --tb=short --durations=5 -k "benchmarked_geometry or bf16_fused_path or one_step_at_224" 2>&1 | tail -40 | cut -c1-600 > $O/r06_parity_b256.log
tail -12 $O/r06_parity_b256.log
cat $O/r06_parity_report_call1.txt | cut -c1-700
# (the ping-pong kernel's row-1 check ran here: profiles/r06_pp_bench_row1.txt; the kernel and its bench tool were deleted after call 2)
timeout 600 python tools/gpu/fwd1x1_layers.py > $O/r06_fwd1x1_layers_before.txt 2>&1; tail -4 $O/r06_fwd1x1_layers_before.txt
timeout 600 python tools/gpu/wrw_layers.py > $O/r06_wrw_layers_before.txt 2>&1; tail -30 $O/r06_wrw_layers_before.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $O/r06_bench_before.json 2> $O/r06_bench_before.err; cut -c1-400 $O/r06_bench_before.json; tail -3 $O/r06_bench_before.err
