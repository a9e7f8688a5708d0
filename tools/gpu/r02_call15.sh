# round 2, GPU call 15: host-bound first process, third attempt -- real HOME; MIOpen's own dirs removed between runs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PF_BENCH_TRACE_STEPS=1
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 4 --no_cpu_baseline > gpurun_out/r02_c15_$name.log 2>&1
  echo "$name: $(grep '"metric"' gpurun_out/r02_c15_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],2))")  $(grep 'host ms/step' gpurun_out/r02_c15_$name.log | cut -c1-200)"
}
wipe() { rm -rf /root/.config/miopen /root/.cache/miopen; }
echo "HOME=$HOME"; ls -la /root | head -20
run first
run second
wipe; run wiped
wipe; run wiped_userdb MIOPEN_USER_DB_PATH=/tmp/udb_only
wipe; run wiped_cachedir MIOPEN_CUSTOM_CACHE_DIR=/tmp/cache_only
wipe; run wiped_nobenchmark PF_CUDNN_BENCHMARK=0
wipe; timeout 600 python -m cProfile -s tottime bench.py --steps 10 --warmup 4 --no_cpu_baseline 2>&1 | grep -v '"metric"' | head -30 | cut -c1-200 > gpurun_out/r02_c15_cprofile.log
head -24 gpurun_out/r02_c15_cprofile.log
