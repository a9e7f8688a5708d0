# Round 6, GPU call 7: the whole GPU suite at HEAD (regression after the round's kernel work)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
rm -f $O/r06_parity_report_mid.txt
PF_PARITY_REPORT=$GRAFT_REPO_ROOT/$O/r06_parity_report_mid.txt timeout 3000 python -m pytest tests -m gpu -q --timeout=1800 --tb=short --durations=6 2>&1 | tail -40 | cut -c1-400 > $O/r06_pytest_gpu_mid.log
tail -30 $O/r06_pytest_gpu_mid.log
timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | cut -c1-300
