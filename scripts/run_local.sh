#!/bin/bash
# Local launcher with the reference's CLI (scripts/run_local.sh:1-48):
#   scripts/run_local.sh <pocketflow_amd/nets/xxx_run.py> [-n=N | --nb_gpus=N] [--flags ...]
# One process per GPU: N == 1 runs the script directly; 1 < N <= 8 launches N ranks with
# torch.distributed.run (RCCL over xGMI) instead of mpirun, adding --enbl_multi_gpu.
nb_gpus=1
py_script="$1"
shift
extra_args=""
for i in "$@"
do
  case "$i" in
    -n=*|--nb_gpus=*)
    nb_gpus="${i#*=}"
    ;;
    *)
    extra_args="${extra_args} ${i}"
    ;;
  esac
done
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "${root}"
conf=path.conf
[ -f "${conf}" ] || conf=path.conf.template
extra_args_path=""
if [ -f "${conf}" ]; then
  extra_args_path=`python -m pocketflow_amd.utils.get_path_args local ${py_script} ${conf}`
fi
extra_args="${extra_args} ${extra_args_path}"
echo "Python script: ${py_script}"
echo "# of GPUs: ${nb_gpus}"
echo "extra arguments: ${extra_args}"

# the first nb_gpus devices (rocm-smi has no "idle GPU" notion comparable to the reference's nvidia-smi scan)
if [ -z "${HIP_VISIBLE_DEVICES}" ]; then
  export HIP_VISIBLE_DEVICES=$(seq -s, 0 $((nb_gpus - 1)))
fi
export HSA_ENABLE_IPC_MODE_LEGACY=0

rm -rf logs && mkdir logs
module=$(echo "${py_script%.py}" | tr '/' '.')
if [ ${nb_gpus} -eq 1 ]; then
  echo "multi-GPU training disabled"
  python -m ${module} ${extra_args}
elif [ ${nb_gpus} -le 8 ]; then
  echo "multi-GPU training enabled"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node ${nb_gpus} --master-addr 127.0.0.1 \
      --master-port ${MASTER_PORT:-29511} -m ${module} --enbl_multi_gpu ${extra_args}
fi
