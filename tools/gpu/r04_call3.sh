# Round 4, third GPU call: where does the recorded-step test die (full logs), and the measured numbers of the new bf16 parity tests.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for t in test_step_graph_ws_resnet20_is_the_eager_step test_step_graph_uq_resnet50_bf16_distillation_is_the_eager_step test_step_graph_cp_mobilenet_is_the_eager_step; do
  timeout 600 python -X faulthandler -m pytest tests/test_learner_gpu.py -m gpu -q -x --tb=short -k "$t" -s > gpurun_out/r04_c3_$t.log 2>&1
  echo "== $t rc=$?"; grep -v amdgpu.ids gpurun_out/r04_c3_$t.log | grep -n "Fatal\|Error\|error\|passed\|failed\|losses eager\|Segmentation\|Abort" | head -12 | cut -c1-300
done
rm -f gpurun_out/r04_c3_parity_report.txt
PF_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r04_c3_parity_report.txt timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -k "bf16" -s > gpurun_out/r04_c3_pytest_bf16_parity.log 2>&1
echo "== bf16 parity rc=$?"; tail -30 gpurun_out/r04_c3_pytest_bf16_parity.log | cut -c1-400
cat gpurun_out/r04_c3_parity_report.txt | cut -c1-420
