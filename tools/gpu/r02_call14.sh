# round 2, GPU call 14: host-bound first process, second attempt -- a fresh HOME per run reproduces "no MIOpen dirs yet"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  h=/tmp/home_$name; rm -rf $h; mkdir -p $h
  env HOME=$h "$@" timeout 600 python bench.py --steps 10 --warmup 4 --no_cpu_baseline > gpurun_out/r02_c14_$name.log 2>&1
  echo "$name: $(grep '"metric"' gpurun_out/r02_c14_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],2))")  files: $(find $h -type f | wc -l)"
}
run default
run nobenchmark PF_CUDNN_BENCHMARK=0
run userdb MIOPEN_USER_DB_PATH=/tmp/udb_only
run cachedir MIOPEN_CUSTOM_CACHE_DIR=/tmp/cache_only
mkdir -p /tmp/home_premade/.config/miopen /tmp/home_premade/.cache/miopen
name=premade; env HOME=/tmp/home_premade timeout 600 python bench.py --steps 10 --warmup 4 --no_cpu_baseline > gpurun_out/r02_c14_$name.log 2>&1
echo "$name: $(grep '"metric"' gpurun_out/r02_c14_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],2))")"
run ownwrw64 PF_OWN_CONV2D_WRW_MIN_C=64
run default_again
# second process on the HOME of the first run
env HOME=/tmp/home_default timeout 600 python bench.py --steps 10 --warmup 4 --no_cpu_baseline 2>&1 | grep '"metric"' | cut -c100-260
