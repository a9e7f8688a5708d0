# Round evidence on the GPU box:  bash tools/gpu/round_evidence.sh [tag]   (tag defaults to r06; outputs gpurun_out/<tag>_*)
# smoke -> full GPU test suite (gradient-parity report lines -> <tag>_parity_report.txt) -> rocprofv3 kernel trace of bench.py
# (stats + steady-state step table with the idle-gap analysis) -> separate PMC passes (FETCH_SIZE / WRITE_SIZE; never combined
# with a trace domain) -> the default bench.py line with cpu_baseline -> one bench line per other configuration of SURVEY 8(d)
# second argument: all (default) | suite (smoke + tests only) | profiles (everything behind the tests) -- two calls when GPU minutes are short
TAG=${1:-r06}
PART=${2:-all}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ $PART != profiles ]; then
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
rm -f gpurun_out/${TAG}_parity_report.txt
PF_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_parity_report.txt timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --tb=line --durations=8 2>&1 | tail -24 | cut -c1-300 > gpurun_out/${TAG}_pytest_gpu.log
cat gpurun_out/${TAG}_pytest_gpu.log
fi
[ $PART = suite ] && exit 0
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no_cpu_baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/${TAG}_prof.log | cut -c1-200
python tools/prof_summary.py $(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/${TAG}_step_kernels_b256.csv | head -14 | cut -c1-150
cp $(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_rocprofv3_stats_b256.csv
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no_cpu_baseline --step_graph 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $c --out $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$c.csv | head -8 | cut -c1-160
done
# MFMA-busy pass (north_star: "rocprof MFMA-busy"): SQ busy cycles of the matrix pipe beside the SQ / GRBM activity counters, one pass, no
# trace domain; tools/mfma_busy.py turns it into a per-kernel table (derivation and calibration in its docstring)
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 5 --no_cpu_baseline --step_graph 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_mfma.log 2>&1
f=$(find /tmp/pmc_mfma -name '*counter_collection.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/pmc_table.py $f > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_mfma_busy.csv 2>> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_mfma.log; head -5 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_mfma_busy.csv | cut -c1-200
# per-kernel table + the whole-step derivation (busy cycles per dispatch x calls per step of the kernel trace above)
python $GRAFT_REPO_ROOT/tools/mfma_busy.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_mfma_busy.csv $GRAFT_REPO_ROOT/gpurun_out/${TAG}_step_kernels_b256.csv > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_mfma_busy.txt 2>&1; tail -7 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_mfma_busy.txt | cut -c1-200
cd $GRAFT_REPO_ROOT
cp gpurun_out/${TAG}_pmc_FETCH_SIZE.csv gpurun_out/${TAG}_pmc_WRITE_SIZE.csv profiles/ 2>/dev/null     # bench.py reads roofline.traffic from profiles/
PF_BENCH_TRACE_STEPS=1 timeout 900 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; grep -v '"metric"' gpurun_out/${TAG}_bench.log | tail -2 | cut -c1-300; grep '"metric"' gpurun_out/${TAG}_bench.log | cut -c1-2200
grep '"metric"' gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
for c in c2a32 c4 c3 c1; do
  timeout 400 python bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err || tail -3 gpurun_out/${TAG}_bench_$c.err
  python -c "
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d = json.loads(ln); print(sys.argv[2], round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms/step', d['config'].get('workload', '')[:90])
" gpurun_out/${TAG}_bench_$c.json $c
done
cd /tmp
# steady-state kernel tables of the other configurations (is any library kernel left in their steps?)
for c in c1 c3; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_${TAG}_$c -o $c -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 12 --warmup 5 --no_cpu_baseline --step_graph 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$c.log 2>&1
  marker=k_adam_flat; [ $c = c1 ] && marker=k_momentum_flat
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/prof_${TAG}_$c -name '*kernel_trace.csv' | head -1) --steps 4 --marker $marker --out $GRAFT_REPO_ROOT/gpurun_out/${TAG}_step_kernels_$c.csv | head -12 | cut -c1-150
done
cd $GRAFT_REPO_ROOT
# per-layer tables at the benchmarked geometry: every 1x1 forward launch of the roofline region, every backward-filter launch, the
# 3x3 forward / backward-data launches against MIOpen, the dense / few-channel kernels against the libraries they replaced
timeout 600 python tools/gpu/fwd1x1_layers.py > gpurun_out/${TAG}_fwd1x1_layers.txt 2>/dev/null; tail -2 gpurun_out/${TAG}_fwd1x1_layers.txt | cut -c1-200
timeout 600 python tools/gpu/wrw_layers.py > gpurun_out/${TAG}_wrw_layers.txt 2>/dev/null; tail -2 gpurun_out/${TAG}_wrw_layers.txt | cut -c1-200
timeout 600 python tools/gpu/igemm_bench.py > gpurun_out/${TAG}_igemm_layers.txt 2>/dev/null; tail -14 gpurun_out/${TAG}_igemm_layers.txt | cut -c1-200
timeout 600 python tools/gpu/new_kernels_bench.py > gpurun_out/${TAG}_new_kernels_layers.txt 2>/dev/null; tail -12 gpurun_out/${TAG}_new_kernels_layers.txt | cut -c1-200
