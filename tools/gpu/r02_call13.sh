# round 2, GPU call 13: host-bound first process -- which MIOpen setting / call causes it?  every run gets an empty MIOpen
# user db + kernel cache (MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR under /tmp)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  d=/tmp/miopen_$name; rm -rf $d; mkdir -p $d/db $d/cache
  env MIOPEN_USER_DB_PATH=$d/db MIOPEN_CUSTOM_CACHE_DIR=$d/cache "$@" timeout 600 python bench.py --steps 10 --warmup 4 --no_cpu_baseline > gpurun_out/r02_c13_$name.log 2>&1
  echo "$name: $(grep '"metric"' gpurun_out/r02_c13_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],2))")"
}
run default
run nobenchmark PF_CUDNN_BENCHMARK=0
run ownwrw64 PF_OWN_CONV2D_WRW_MIN_C=64
run ownwrw64_nobenchmark PF_OWN_CONV2D_WRW_MIN_C=64 PF_CUDNN_BENCHMARK=0
run findmode2 MIOPEN_FIND_MODE=2
d=/tmp/miopen_prof; rm -rf $d; mkdir -p $d/db $d/cache
MIOPEN_USER_DB_PATH=$d/db MIOPEN_CUSTOM_CACHE_DIR=$d/cache timeout 600 python -m cProfile -s tottime bench.py --steps 10 --warmup 4 --no_cpu_baseline 2>&1 | grep -v '"metric"' | head -40 | cut -c1-200 > gpurun_out/r02_c13_cprofile.log
head -30 gpurun_out/r02_c13_cprofile.log
