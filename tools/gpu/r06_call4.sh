# Round 6, GPU call 4: the consumer's inference-mode BN folded into the producing convolution (teacher forward) -- kernel tests, bit
# identity of the teacher's logits, A/B in one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_igemm_gpu.py tests/test_bench_geometry_gpu.py -m gpu -q --tb=short -k "inference_bn or teacher_conv" 2>&1 | tail -8 | cut -c1-300
timeout 600 python - <<'PY' 2>&1 | tail -4
import os, sys, tempfile, torch
sys.path.insert(0, os.getcwd())
sys.argv = ['bench.py', '--batch', '64', '--no_cpu_baseline']
import bench
from pocketflow_amd.flags import FLAGS
from pocketflow_amd import graph as G
args = bench.parse_args()
tmp = tempfile.mkdtemp()
learner, step = bench.build_learner(args, FLAGS, tmp, 0, 1, lambda: None)
x, y = learner.to_device(*learner.iter_train.get_next())
t = learner.helper_dst
G.FOLD_EVAL_BN = False
a = t.calc_logits(None, x).clone()
G.FOLD_EVAL_BN = True
b = t.calc_logits(None, x).clone()
torch.cuda.synchronize()
print('teacher logits with bn2 / bn3 folded into conv1 / conv2: bit-identical %s (max abs diff %.3e, %d x %d logits, std %.3f)' % (
    bool(torch.equal(a, b)), float((a.float() - b.float()).abs().max()), a.shape[0], a.shape[1], float(a.float().std())))
PY
run() {  # label, env...
  label=$1; shift
  v=$(env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); sg = d['config'].get('step_graph'); print('%.0f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], 'recorded' if sg else 'launch by launch'))
")
  echo "$label | $v" | tee -a $O/r06_fold_eval_bn_ab.txt
  [ -z "$v" ] && tail -3 $O/r06_ab_err.txt
}
rm -f $O/r06_fold_eval_bn_ab.txt
echo "# teacher's bn2 / bn3 (inference mode) applied in the epilogue of conv1 / conv2 (PF_FOLD_EVAL_BN), one box, bench.py --steps 20 --warmup 5 --no_cpu_baseline" >> $O/r06_fold_eval_bn_ab.txt
run "stand-alone passes (PF_FOLD_EVAL_BN=0)   " PF_FOLD_EVAL_BN=0
run "folded (default)                         " PF_X=0
run "stand-alone passes (PF_FOLD_EVAL_BN=0)   " PF_FOLD_EVAL_BN=0
run "folded (default)                         " PF_X=0
for c in c4 c2a32; do
  for f in 0 1; do
    v=$(PF_FOLD_EVAL_BN=$f timeout 400 python bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline 2>$O/r06_ab_err.txt | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('%.0f images/s  %.2f ms/step' % (d['value'], d['ms_per_step']))
")
    echo "$c PF_FOLD_EVAL_BN=$f | $v" | tee -a $O/r06_fold_eval_bn_ab.txt
  done
done
