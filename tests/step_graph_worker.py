"""The recorded-step cases of tests/test_learner_gpu.py, run in a process of their own (python tests/step_graph_worker.py <case> <tmp dir>).

Why a worker process: a crash inside the HIP runtime's graph instantiation (round 4: hipStreamEndCapture died with SIGSEGV whenever the
PREVIOUS step's autograd graph was still alive while a step was recorded -- the learners now return detached tensors) must cost one
test, not the rest of the GPU suite.  Prints one JSON line: {"case", "exact", "losses_eager", "losses_graph", ...}.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathlib

import numpy as np
import torch


def _setup(tmp_path, **kw):
  from pocketflow_amd.flags import FLAGS
  import pocketflow_amd.learners.learner_utils  # noqa: F401  (defines flags)
  import pocketflow_amd.learners.abstract_learner  # noqa: F401
  import pocketflow_amd.datasets.abstract_dataset  # noqa: F401  (synthetic_pool)
  FLAGS.save_path = str(tmp_path / 'models' / 'model.ckpt')
  FLAGS.save_path_eval = str(tmp_path / 'models_eval' / 'model.ckpt')
  FLAGS.synthetic_pool = 2
  for k, v in kw.items():
    setattr(FLAGS, k, v)
  return FLAGS


RESULT = {}


SIGNATURES = {}


def _signature(lrn):
  """Diagnostics (PF_W_DIAG=1): checksums of the reduced gradient buckets and of the flat buffers after a step."""
  st = lrn.graph.store
  red = getattr(st, 'reducer', None)
  sig = {'w_master': float(st.w_master.double().abs().sum()), 'o_master': float(st.o_master.double().abs().sum()),
         'state': float(st.state.double().abs().sum())}
  if red is not None and red._stage_w is not None:
    sig['buckets'] = [[float(red._stage_w[lo:hi].double().sum()), float(red._stage_w[lo:hi].double().abs().sum())] for lo, hi, _ in red.buckets]
    if red._stage_o is not None:
      sig['o_stage'] = float(red._stage_o.double().abs().sum())
    sig['vars'] = {v.name: float(red._stage_w[v.offset:v.offset + v.numel].double().abs().sum()) for v in st.vars if v.group == 'W' and v.trainable}
    sig['local'] = dict(getattr(red, '_diag_local', {}))
  return sig


def _collect_losses_each_step(lrn, n_steps, suspend_at, graph_mode):
  from pocketflow_amd import step_graph
  losses = []
  diag = os.environ.get('PF_W_DIAG') == '1'
  red0 = getattr(lrn.graph.store, 'reducer', None)
  if diag and red0 is not None:
    fin0 = red0.finish

    def fin():
      if not lrn.graph.capturing:
        st = lrn.graph.store
        red0._diag_local = {v.name: float(st.w_grad[v.offset:v.offset + v.numel].double().abs().sum()) for v in st.vars if v.group == 'W' and v.trainable}
      return fin0()
    red0.finish = fin
  for i in range(n_steps):
    if graph_mode and i in suspend_at:
      sg = step_graph.of(lrn)
      sg.resume() if sg.suspended else sg.suspend()
    o = lrn.train_step()
    if diag:
      torch.cuda.synchronize()
      SIGNATURES.setdefault('graph' if graph_mode else 'eager', []).append(_signature(lrn))
    losses.append((o['loss'] if isinstance(o, dict) else o[1]).detach().clone())
    # (`o` stays bound while the next step runs, as in the learners' train() loops: the returned tensors carry no autograd graph)
  return [float(l) for l in losses]


def _assert_same_run(a, b, la, lb, what, loss_rtol=1e-4, param_atol=2e-3):
  """Eager run `a` vs recorded run `b`: same batches in the same order.  Every kernel of this library is deterministic, so the two
  runs are bit-identical wherever all launches are ours; a step that still contains library convolutions (MIOpen picks its
  solver differently under stream capture: measured 1e-5 relative on a ResNet-20 loss) is held to `loss_rtol` / `param_atol`.
  Whether the run WAS bit-identical is printed."""
  sa, sb = a.graph.store, b.graph.store
  if SIGNATURES:
    for i, (e, g) in enumerate(zip(SIGNATURES['eager'], SIGNATURES['graph'])):
      print('DIAG rank %s step %d loss %.9g | %.9g %s' % (os.environ.get('RANK', '0'), i, la[i], lb[i], 'SAME' if e == g else 'DIFFERENT'))
      if e != g:
        ev, gv, el, gl = e.pop('vars', {}), g.pop('vars', {}), e.pop('local', {}), g.pop('local', {})
        print('DIAG   eager %s' % json.dumps(e))
        print('DIAG   graph %s' % json.dumps(g))
        bad = [k for k in ev if ev[k] != gv.get(k)]
        print('DIAG   reduced gradients that differ: %d of %d: %s' % (len(bad), len(ev), ' '.join('%s(%.3e)' % (k.split('/', 1)[-1], abs(ev[k] - gv[k]) / max(abs(ev[k]), 1e-30)) for k in bad[:12])))
        badl = [k for k in el if el[k] != gl.get(k)]
        print('DIAG   local gradients that differ: %d of %d: %s' % (len(badl), len(el), ' '.join('%s(%.3e)' % (k.split('/', 1)[-1], abs(el[k] - gl[k]) / max(abs(el[k]), 1e-30)) for k in badl[:12])))
  worst = max(float((x - y).abs().max()) for x, y in ((sa.w_master, sb.w_master), (sa.o_master, sb.o_master), (sa.state, sb.state)))
  exact = la == lb and worst == 0.0
  RESULT.update(what=what, exact=bool(exact), losses_eager=la, losses_graph=lb, max_parameter_difference=worst)
  assert all(abs(x - y) <= loss_rtol * max(1.0, abs(x)) for x, y in zip(la, lb)), (what, la, lb)
  assert worst <= param_atol, (what, worst)
  return exact


def case_uq_resnet50(tmp_path, ranks=1, batch=8, image=64):
  """BASELINE configs[2] shrunk (ResNet-v2-50 @64, batch 8, UQ w8/a8 + distillation, bf16 fused path): 9 steps launch by launch
  vs 3 eager + recording + replays with the graph suspended for steps 6-7.  The teacher's forward over the next batch is a forked
  branch of the graph; Adam's alpha_t comes from device memory.  Same batches in the same order, deterministic kernels: the losses
  and every parameter, Adam slot and BN moving statistic must be bit-identical."""
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd.nets.resnet_at_ilsvrc12 import ModelHelper
  from pocketflow_amd.learners.uniform_quantization.learner import UniformQuantLearner
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  from pocketflow_amd import step_graph
  import pocketflow_amd.learners.distillation_helper  # noqa: F401
  _setup(tmp_path, batch_size=batch, batch_size_eval=batch, uql_weight_bits=8, uql_activation_bits=8, enbl_dst=True, dst_eval_teacher=False,
         save_path_dst=str(tmp_path / 'models_dst' / 'model.ckpt'), uql_save_quant_model_path=str(tmp_path / 'uql' / 'm.ckpt'),
         resnet_size=50, nb_classes=1001, image_size=image, compute_dtype='bfloat16', synthetic_pool=int(os.environ.get('PF_W_POOL', '5')))
  torch.backends.cudnn.benchmark = os.environ.get('PF_W_BENCHMARK', '0') != '0'
  if os.environ.get('PF_W_STRICT', '1') != '0':
    os.environ['PF_STEP_GRAPH_STRICT'] = '1'
  made = []

  def make():
    mh = ModelHelper()
    if not made:
      if ranks == 1 or torch.distributed.get_rank() == 0:
        create_synthetic_checkpoint(mh)                      # (several ranks: one scratch directory, written by rank 0)
      if ranks > 1:
        torch.distributed.barrier()
    made.append(1)
    return UniformQuantLearner(None, mh)
  # (the recorded run first: round 4 saw hipStreamEndCapture crash when ANOTHER learner with live side-stream work existed in the
  # process -- see DESIGN.md; a process normally owns one learner)
  FLAGS.enbl_step_graph = True
  b = make()
  lb = _collect_losses_each_step(b, 9, (6, 8), True)
  FLAGS.enbl_step_graph = False
  a = make()
  import pocketflow_amd.graph as G
  forks_b = getattr(b.graph.store, 'wrw_side', None)
  if os.environ.get('PF_W_ONE_QUEUE_EAGER') == '1':
    # the launch-by-launch run with the backward-filter launches in the ONE queue (graph.WrwSide off): the recorded run above forked them
    assert G.WRW_SIDE and forks_b is not None, 'the recorded run did not fork its backward-filter launches'
    G.WRW_SIDE = False
  la = _collect_losses_each_step(a, 9, (), False)
  if os.environ.get('PF_W_ONE_QUEUE_EAGER') == '1':
    assert getattr(a.graph.store, 'wrw_side', None) is None
    G.WRW_SIDE = True
  sg = step_graph.of(b)
  assert sg.state == 'ready' and sg.error is None and sg.n_replays == 9 - 3 - 2 and sg.nxt is not None
  exact = _assert_same_run(a, b, la, lb, 'ResNet-50 UQ bf16 + dst')
  assert a.optimizer.slots_w[1].abs().sum() > 0
  assert not exact or torch.equal(a.optimizer.slots_w[1], b.optimizer.slots_w[1])


def case_ws_resnet20(tmp_path):
  """BASELINE configs[1] (ResNet-20 @ CIFAR-10, WeightSparseLearner, bf16): no teacher -> single-stream graph; Momentum's learning
  rate from device memory; masks at a non-trivial ratio applied inside the recorded optimiser launch."""
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd.nets.resnet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.weight_sparsification.learner import WeightSparseLearner
  from pocketflow_amd import step_graph
  _setup(tmp_path, batch_size=32, batch_size_eval=32, ws_prune_ratio=0.5, ws_prune_ratio_prtl='uniform',
         ws_save_path=str(tmp_path / 'ws' / 'm.ckpt'), resnet_size=20, nb_classes=10, ws_mask_update_step=2,
         nb_smpls_train=32 * 12, nb_epochs_rat=1.0 / 250, compute_dtype='bfloat16', synthetic_pool=5)
  os.environ['PF_STEP_GRAPH_STRICT'] = '1'

  def run(graph_mode):
    FLAGS.enbl_step_graph = graph_mode
    lrn = WeightSparseLearner(None, ModelHelper())
    lrn.global_step = int(0.3 * lrn.nb_iters_train)
    lrn.prune_step()                                         # masks at a non-trivial ratio BEFORE the steps (identical on both sides)
    lrn.global_step = 0
    losses = []
    for it in range(9):
      losses.append(lrn.train_step()[1].detach().clone())
    return lrn, [float(l) for l in losses]
  b, lb = run(True)
  a, la = run(False)
  sg = step_graph.of(b)
  assert sg.state == 'ready' and sg.error is None and sg.n_replays == 6 and sg.nxt is None
  assert torch.equal(a.masks, b.masks) and 0.05 < 1.0 - float(a.masks.mean()) < 0.6
  _assert_same_run(a, b, la, lb, 'ResNet-20 WS bf16')
  st = b.graph.store
  for v in b.maskable_vars:
    m = b.masks[v.offset:v.offset + v.numel]
    assert float((st.w_master[v.offset:v.offset + v.numel] * (1 - m)).abs().max()) == 0.0


def case_cp_mobilenet(tmp_path):
  """BASELINE configs[3] shrunk (MobileNet-v1 x0.5 @64, channel-pruned masked fine-tune + distillation, bf16): the dropout mask of
  every step reaches the recorded launches through graph.step_feeders."""
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd import step_graph
  import pocketflow_amd.learners.distillation_helper  # noqa: F401
  from pocketflow_amd.nets.mobilenet_at_ilsvrc12 import ModelHelper
  from pocketflow_amd.learners.channel_pruning.learner import ChannelPrunedLearner
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  from pocketflow_amd.utils import checkpoint
  _setup(tmp_path, batch_size=16, batch_size_eval=16, image_size=64, nb_classes=101, mobilenet_depth_mult=0.5, enbl_dst=True,
         dst_eval_teacher=False, save_path_dst=str(tmp_path / 'models_dst' / 'model.ckpt'), compute_dtype='bfloat16', synthetic_pool=5,
         cp_channel_pruned_path=str(tmp_path / 'models' / 'pruned_model.ckpt'), cp_best_path=str(tmp_path / 'models' / 'best_model.ckpt'),
         cp_original_path=str(tmp_path / 'models' / 'original_model.ckpt'), cp_lrn_rate_ft=1e-4)
  os.environ['PF_STEP_GRAPH_STRICT'] = '1'
  made = []

  def make():
    mh = ModelHelper()
    if not made:
      create_synthetic_checkpoint(mh)
    made.append(1)
    lrn = ChannelPrunedLearner(None, mh)
    rng = np.random.RandomState(11)
    convs = [op for op in lrn.graph.matmul_ops if op.var.kind == 'conv']
    vals = lrn.graph.store.export_numpy()
    fake = {}
    for i, op in enumerate(convs):
      kh, kw, cin, cout = op.var.ref_shape
      keep_in = np.ones(cin, bool) if i == 0 else rng.rand(cin) < 0.5
      keep_out = np.ones(cout, bool) if i == len(convs) - 1 else rng.rand(cout) < 0.5
      keep_in[0] = keep_out[0] = True
      fake[op.name] = [keep_in.tolist(), keep_out.tolist()]
      m = np.zeros(op.var.ref_shape, np.float32)
      m[:, :, keep_in, :] = 1.0
      m[:, :, :, ~keep_out] = 0.0
      vals[op.var.name] = vals[op.var.name] * m
    pruned = checkpoint.save(vals, FLAGS.cp_channel_pruned_path, None)
    lrn.setup_finetune(pruned, finetune=True, fake_pruning_dict=fake)
    net = lrn.graph.nets['mobilenet']
    net.keep = 0.8                                          # (the default 0.999 leaves almost every mask all-ones)
    return lrn
  FLAGS.enbl_step_graph = True
  b = make()
  lb = _collect_losses_each_step(b, 8, (), True)
  FLAGS.enbl_step_graph = False
  a = make()
  la = _collect_losses_each_step(a, 8, (), False)
  sg = step_graph.of(b)
  assert sg.state == 'ready' and sg.error is None and sg.n_replays == 5
  assert a.graph.nets['mobilenet'].dropout_step == b.graph.nets['mobilenet'].dropout_step == 8
  _assert_same_run(a, b, la, lb, 'MobileNet-v1 CP bf16 + dst')
  for op in b.graph.matmul_ops:
    if op.name in b.fake_pruning_dict and op.var.kind == 'conv':
      keep_in, keep_out = [np.asarray(k, bool) for k in b.fake_pruning_dict[op.name]]
      w = op.var.to_ref(op.var.master.detach().cpu().numpy())
      assert np.all(w[:, :, ~keep_in, :] == 0) and np.all(w[:, :, :, ~keep_out] == 0), op.name


def case_uq_resnet50_two_ranks(tmp_path):
  """case_uq_resnet50 under torch.distributed.run with TWO ranks on one GPU (PF_DIST_BACKEND=gloo, PF_SINGLE_DEVICE=1: RCCL refuses
  duplicate devices): --enbl_multi_gpu, so the recorded step is TWO hipGraphs around the gradient exchange (step_graph.CudaBackend.cut,
  made by optim.GradReducer.finish() behind the captured backward pass).  At 224 x 224 with 48 images per rank every launch of the
  step is a kernel of this library (at 64 x 64 some backward-filter shapes fall back to MIOpen, whose choice depends on the
  allocator's state: two launch-by-launch runs then differ in a few gradients by 1e-8 .. 3e-5, which Adam amplifies), so recorded
  and launch-by-launch runs must agree bit for bit on each rank, and the ranks stay in lock-step."""
  import torch.distributed as dist
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd import step_graph
  from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw
  import pocketflow_amd.learners.abstract_learner  # noqa: F401
  FLAGS.enbl_multi_gpu = True
  mgw.init()
  rank = mgw.rank()
  os.environ.setdefault('PF_ALLREDUCE_BUCKET', str(6 << 20))
  made = {}
  orig = step_graph.StepGraph._record

  def spy(self):
    orig(self)
    made['graphs'], made['actions'] = len(self.backend.graphs), len(self.backend.actions)
  step_graph.StepGraph._record = spy
  case_uq_resnet50(tmp_path, ranks=2, batch=int(os.environ.get('PF_W_BATCH', '48')), image=int(os.environ.get('PF_W_IMAGE', '224')))
  assert made['graphs'] == 2 and made['actions'] == 1, made
  RESULT.update(rank=rank, **made)
  sig = torch.tensor([RESULT['losses_graph'][-1]], dtype=torch.float64)
  both = [torch.zeros_like(sig) for _ in range(2)]
  dist.all_gather(both, sig)
  assert float(both[0]) != float(both[1]), both                # per-rank data ...
  RESULT['last_loss_of_each_rank'] = [float(t) for t in both]
  dist.barrier()
  if rank != 0:
    RESULT.clear()                                             # one result line (rank 0's)
  dist.destroy_process_group()


CASES = {'uq_resnet50': case_uq_resnet50, 'ws_resnet20': case_ws_resnet20, 'cp_mobilenet': case_cp_mobilenet,
         'uq_resnet50_two_ranks': case_uq_resnet50_two_ranks}

if __name__ == '__main__':
  case, tmp = sys.argv[1], pathlib.Path(sys.argv[2])
  CASES[case](tmp)
  if RESULT:
    RESULT['case'] = case
    print('STEP_GRAPH_RESULT ' + json.dumps(RESULT))
