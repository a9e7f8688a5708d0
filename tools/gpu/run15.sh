cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_learner_gpu.py -m gpu -q --timeout=300 --tb=line -k "fused_path or mobilenet or resnet20" 2>&1 | tail -4 | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 5 --no_cpu_baseline 2>&1 | tail -1 | cut -c1-1400
