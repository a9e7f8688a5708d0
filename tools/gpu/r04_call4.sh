# Round 4, fourth GPU call: (1) the bf16 parity tests' measured numbers; (2) where hipStreamEndCapture dies at small shapes.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r04_c4_parity_report.txt
PF_PARITY_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r04_c4_parity_report.txt timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -k "bf16" -s > gpurun_out/r04_c4_pytest_bf16_parity.log 2>&1
echo "== bf16 parity rc=$?"; tail -40 gpurun_out/r04_c4_pytest_bf16_parity.log | grep -v "^  File\|Warning" | cut -c1-400
cat gpurun_out/r04_c4_parity_report.txt | cut -c1-420
probe() { tag=$1; shift; timeout 300 "$@" > gpurun_out/r04_c4_probe_$tag.log 2>&1; echo "probe $tag rc=$? $(grep -c 'Segmentation' gpurun_out/r04_c4_probe_$tag.log) $(grep -o '"value": [0-9.]*' gpurun_out/r04_c4_probe_$tag.log | head -1) $(grep -i 'not recorded\|recording failed' gpurun_out/r04_c4_probe_$tag.log | head -1 | cut -c1-200)"; }
B="python bench.py --steps 6 --warmup 5 --no_cpu_baseline --step_graph 1"
probe A_c2_b8_64 $B --config c2 --batch 8 --image_size 64
PF_CUDNN_BENCHMARK=0 probe B_c2_b8_64_nobench $B --config c2 --batch 8 --image_size 64
PF_TEACHER_AHEAD=0 probe C_c2_b8_64_noahead $B --config c2 --batch 8 --image_size 64
probe D_c3_b16_64 $B --config c3 --batch 16 --image_size 64
probe E_c2_b32_128 $B --config c2 --batch 32 --image_size 128
probe F_c2a32_b8_64 $B --config c2a32 --batch 8 --image_size 64
PF_TEST_POISON=0 probe G_pytest_uq_nopoison python -X faulthandler -m pytest tests/test_learner_gpu.py -m gpu -q -x --tb=short -k test_step_graph_uq_resnet50 -s
