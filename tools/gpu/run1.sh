set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
nproc; rocm-smi --showproductname 2>/dev/null | head -5
python -c "import torch; print(torch.cuda.get_device_name(0))"
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -40 > gpurun_out/pytest1.log
tail -40 gpurun_out/pytest1.log
timeout 600 python bench.py --steps 5 --warmup 3 --batch 64 --no_cpu_baseline > gpurun_out/bench_b64.log 2>&1
tail -5 gpurun_out/bench_b64.log
