# round 2, GPU call 26: probe-and-restart of bench.py (forced once); kernel trace of the final code; default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
PF_BENCH_FORCE_RESTART=1 timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline 2>gpurun_out/r02_c26_restart.err | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('forced restart:', round(d['value']), d['launch_probe'])"
grep -i "restarting" gpurun_out/r02_c26_restart.err | cut -c1-200
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c26 -o c26 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --batch 256 --no_cpu_baseline --no_reexec > $GRAFT_REPO_ROOT/gpurun_out/r02_c26_prof.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/r02_c26_prof.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('under rocprofv3:', round(d['value']), round(d['ms_per_step'],2), d['launch_probe'])"
python tools/prof_summary.py $(find /tmp/prof_c26 -name '*kernel_trace.csv' | head -1) --steps 4 --out gpurun_out/r02_c26_step_kernels_b256.csv | head -5 | cut -c1-150
cp $(find /tmp/prof_c26 -name '*kernel_stats.csv' | head -1) gpurun_out/r02_c26_rocprofv3_stats_b256.csv
timeout 600 python bench.py --no_cpu_baseline 2>/dev/null | grep '"metric"' > gpurun_out/r02_c26_bench.json; python -c "import json; d=json.load(open('gpurun_out/r02_c26_bench.json')); print('default:', round(d['value']), round(d['ms_per_step'],2), d['launch_probe'])"
