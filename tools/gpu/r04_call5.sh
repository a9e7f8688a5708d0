# Round 4, fifth GPU call: general convolution kernels (pf_convg), strided backward-data by parity classes, recorded-step tests
# (recorded run first), recalibrated bf16 parity tests, float32 parity through the in-tree kernels, bench.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { tag=$1; shift; timeout 1200 "$@" > gpurun_out/r04_c5_$tag.log 2>&1; echo "== $tag rc=$?"; grep -v "amdgpu.ids\|Warning\|warnings.warn" gpurun_out/r04_c5_$tag.log | tail -${TAILN:-6} | cut -c1-330; }
run convg python -m pytest tests/test_convg_gpu.py -m gpu -q -x --tb=short
run igemm python -m pytest tests/test_igemm_gpu.py tests/test_conv_gpu.py -m gpu -q --tb=line
TAILN=12 run step_graph python -X faulthandler -m pytest tests/test_learner_gpu.py -m gpu -q --tb=short -k step_graph -s
rm -f gpurun_out/r04_c5_parity_report.txt
export PF_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r04_c5_parity_report.txt
TAILN=14 run bf16_parity python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -k "bf16 and (fused_path or ws_ or cp_)"
TAILN=14 run f32_parity python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -k "lenet or resnet20_distillation or cp_mobilenet_masked or float32_gradients"
cat gpurun_out/r04_c5_parity_report.txt | cut -c1-400
unset PF_PARITY_REPORT
for v in "" "PF_OWN_CONV2D_BWD_STRIDED=0 PF_OWN_CONV_GENERIC=0"; do
  env $v timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r04_c5_bench_c2_$(echo $v | tr -c 'A-Z0-9' '_').json 2> gpurun_out/r04_c5_bench.err
  python -c "
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        d = json.loads(ln); print(sys.argv[2] or 'own', round(d['value']), 'img/s', round(d['ms_per_step'], 2), 'ms/step | roofline frac', round(d['roofline']['frac'], 4))
" gpurun_out/r04_c5_bench_c2_$(echo $v | tr -c 'A-Z0-9' '_').json "$v"
done
