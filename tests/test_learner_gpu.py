"""End-to-end learner steps on the GPU (smoke + invariants); step-level parity against the CPU
oracle learner lives in tests/test_parity_gpu.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_uq_lenet_steps(tmp_path, dtype):
  from pocketflow_amd.nets.lenet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.uniform_quantization.learner import UniformQuantLearner
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  FLAGS = _setup(tmp_path, batch_size=32, uql_weight_bits=8, uql_activation_bits=8, compute_dtype=dtype,
                 uql_save_quant_model_path=str(tmp_path / 'uql' / 'm.ckpt'), nb_eval_batches_override=2)
  mh = ModelHelper()
  create_synthetic_checkpoint(mh)
  learner = UniformQuantLearner(None, mh)
  assert learner.statistics['nb_matmuls'] == 2 and learner.statistics['nb_activations'] == 3
  w0 = learner.graph.store.w_master.clone()
  for _ in range(3):
    out = learner.train_step()
    assert np.isfinite(float(out['loss']))
  assert not torch.equal(w0, learner.graph.store.w_master)
  # quantised compute copy of conv2 holds at most 256 distinct values; conv1 (first layer) is a plain cast
  st = learner.graph.store
  v = st.by_name['model/conv2/kernel']
  assert torch.unique(st.w_compute[v.offset:v.offset + v.numel].float()).numel() <= 256
  v1 = st.by_name['model/conv1/kernel']
  torch.testing.assert_close(st.w_compute[v1.offset:v1.offset + v1.numel].float(),
                             st.w_master[v1.offset:v1.offset + v1.numel], rtol=1e-2, atol=1e-2)


def test_uq_resnet20_with_distillation_and_eval(tmp_path):
  from pocketflow_amd.nets.resnet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.uniform_quantization.learner import UniformQuantLearner
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  FLAGS = _setup(tmp_path, batch_size=16, batch_size_eval=16, uql_weight_bits=8, uql_activation_bits=8,
                 enbl_dst=True, save_path_dst=str(tmp_path / 'models_dst' / 'model.ckpt'),
                 uql_save_quant_model_path=str(tmp_path / 'uql' / 'm.ckpt'), nb_eval_batches_override=2,
                 nb_iters_override=4, summ_step=2, resnet_size=20, nb_classes=10)
  mh = ModelHelper()
  create_synthetic_checkpoint(mh)
  learner = UniformQuantLearner(None, mh)
  assert learner.statistics['nb_matmuls'] == 21 and learner.statistics['nb_activations'] == 19
  rslt = learner.train()
  assert np.isfinite(rslt['loss'])
  # teacher == initial student, so at step 0 the distillation gradient is ~0 only through quant noise
  assert os.path.exists(str(tmp_path / 'uql' / 'checkpoint'))


def test_full_prec_resnet20_two_steps(tmp_path):
  from pocketflow_amd.nets.resnet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.full_precision.learner import FullPrecLearner
  FLAGS = _setup(tmp_path, batch_size=16, resnet_size=20, nb_classes=10)
  learner = FullPrecLearner(None, ModelHelper())
  l0 = float(learner.train_step()[1])
  for _ in range(5):
    l = float(learner.train_step()[1])
  assert np.isfinite(l) and l < l0 * 1.5


@pytest.mark.parametrize('use_buckets,opt_mode', [(False, 'weights'), (True, 'both')])
def test_nuq_resnet20_steps_and_eval(tmp_path, use_buckets, opt_mode):
  from pocketflow_amd.nets.resnet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.nonuniform_quantization.learner import NonUniformQuantLearner
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  FLAGS = _setup(tmp_path, batch_size=16, batch_size_eval=16, nuql_weight_bits=4, nuql_use_buckets=use_buckets,
                 nuql_opt_mode=opt_mode, nuql_save_quant_model_path=str(tmp_path / 'nuql' / 'm.ckpt'),
                 nb_eval_batches_override=2, nb_iters_override=3, summ_step=2, resnet_size=20, nb_classes=10)
  mh = ModelHelper()
  create_synthetic_checkpoint(mh)
  learner = NonUniformQuantLearner(None, mh)
  rslt = learner.train()
  assert np.isfinite(rslt['loss'])
  # every quantised kernel holds at most 2**4 distinct values per bucket; check the per-tensor case
  if not use_buckets:
    st = learner.graph.store
    v = st.by_name['model/resnet_model/conv2d_3/kernel']
    assert torch.unique(st.w_compute[v.offset:v.offset + v.numel].float()).numel() <= 16


def test_ws_resnet20_masks_and_sparsity(tmp_path):
  from pocketflow_amd.nets.resnet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.weight_sparsification.learner import WeightSparseLearner
  FLAGS = _setup(tmp_path, batch_size=16, batch_size_eval=16, ws_prune_ratio=0.5, ws_prune_ratio_prtl='uniform',
                 ws_save_path=str(tmp_path / 'ws' / 'm.ckpt'), nb_eval_batches_override=2, resnet_size=20,
                 nb_classes=10, ws_mask_update_step=2, summ_step=4, nb_smpls_train=16 * 20, nb_epochs_rat=1.0 / 250)
  learner = WeightSparseLearner(None, ModelHelper())
  assert learner.nb_iters_train == 20
  rslt = learner.train()
  assert np.isfinite(rslt['loss'])
  # after the last refresh (step >= 0.5 * N) every maskable tensor sits at the final ratio
  st = learner.graph.store
  for v in learner.maskable_vars:
    w = st.w_master[v.offset:v.offset + v.numel]
    sparsity = float((w == 0).float().mean())
    assert abs(sparsity - 0.5) < 2.0 / v.numel + 1e-3, (v.name, sparsity)
  assert abs(rslt['pr_msk'] - 0.5) < 1e-2


def test_channel_pruned_mobilenet_uniform(tmp_path):
  """ChannelPrunedLearner on MobileNet-v1 (BASELINE config 3, shrunk): LASSO pruning on rank 0, then the
  masked fine-tune -- pruned input / output channels stay exactly zero, FLOPs drop as requested."""
  from pocketflow_amd.nets.mobilenet_at_ilsvrc12 import ModelHelper
  from pocketflow_amd.learners.channel_pruning.learner import ChannelPrunedLearner
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  FLAGS = _setup(tmp_path, batch_size=16, batch_size_eval=16, image_size=64, nb_classes=17, mobilenet_depth_mult=0.5,
                 cp_prune_option='uniform', cp_uniform_preserve_ratio=0.5, cp_nb_batches=8, cp_nb_points_per_layer=10,
                 cp_channel_pruned_path=str(tmp_path / 'models' / 'pruned_model.ckpt'),
                 cp_best_path=str(tmp_path / 'models' / 'best_model.ckpt'),
                 cp_original_path=str(tmp_path / 'models' / 'original_model.ckpt'),
                 nb_eval_batches_override=2, nb_iters_override=3, summ_step=2, synthetic_pool=8)
  mh = ModelHelper()
  create_synthetic_checkpoint(mh)
  learner = ChannelPrunedLearner(None, mh)
  rslt = learner.train()
  assert np.isfinite(rslt['loss'])
  pr = learner.pruner
  assert 0.2 < pr.preserve_ratio < 0.6                                  # ~0.5^2 on the pointwise convs + unpruned ends
  st = learner.graph.store
  n_masked = 0
  for op in learner.graph.matmul_ops:
    if op.name not in learner.fake_pruning_dict or op.var.kind != 'conv':
      continue
    keep_in, keep_out = [np.asarray(k, bool) for k in learner.fake_pruning_dict[op.name]]
    w = op.var.to_ref(op.var.master.detach().cpu().numpy())
    assert np.all(w[:, :, ~keep_in, :] == 0) and np.all(w[:, :, :, ~keep_out] == 0), op.name
    n_masked += int((~keep_in).sum() + (~keep_out).sum())
  assert n_masked > 0
  first, last = learner.pruner.thisconvs[0], learner.pruner.thisconvs[-1]
  assert all(learner.fake_pruning_dict[first.op.name][0]) and all(learner.fake_pruning_dict[last.op.name][1])


@pytest.mark.parametrize('model', ['resnet', 'mobilenet'])
def test_cp_feature_sampling_matches_oracle(tmp_path, model):
  """sampled features / convolution inputs / residual diffs of the channel pruner in the reference's
  setting (own points per tensor name, batch-statistics BN) against oracle/cp_features_oracle.py on the oracle network
  (float32 compute: the pruner is a float32 procedure in the reference).  Body: tests/parity_common.py."""
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  import pocketflow_amd.nets.mobilenet_at_ilsvrc12  # noqa: F401  (flag definitions)
  import pocketflow_amd.nets.resnet_at_cifar10  # noqa: F401
  import pocketflow_amd.learners.channel_pruning.learner  # noqa: F401
  from parity_common import run_cp_feature_sampling_parity
  FLAGS = _setup(tmp_path, compute_dtype='float32')
  r = run_cp_feature_sampling_parity(FLAGS, tmp_path, model)
  print('a17 %s: %d convolutions, %d residual sums, worst relative error %.2e' % (model, r['convs'], r['adds'], r['worst']))
  assert r['convs'] == (10 if model == 'resnet' else 15) and r['adds'] == (6 if model == 'resnet' else 0), r


@pytest.mark.parametrize('model,use_buckets,bucket_type,bits', [('lenet', True, 'split', 3), ('resnet', True, 'channel', 4), ('resnet', False, 'channel', 8)])
def test_int_export_of_a_uq_learner(tmp_path, model, use_buckets, bucket_type, bits):
  """integer codes read back from the device quantiser (pf_seg_minmax / pf_seg_uq_apply) decode bit for bit to
  the oracle's fake-quantised weights; a learner restored from the artefact multiplies with the same numbers.  Body: tests/
  parity_common.py."""
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  import pocketflow_amd.nets.lenet_at_cifar10  # noqa: F401  (flag definitions)
  import pocketflow_amd.nets.resnet_at_cifar10  # noqa: F401
  import pocketflow_amd.learners.uniform_quantization.learner  # noqa: F401
  from parity_common import run_int_export_roundtrip
  FLAGS = _setup(tmp_path)
  summ = run_int_export_roundtrip(FLAGS, tmp_path, model, use_buckets, bucket_type, bits)
  assert summ['quantised_tensors'] == (2 if model == 'lenet' else 21)


@pytest.mark.parametrize('recorded', [False, True])
def test_bench_two_ranks_share_one_gpu(tmp_path, recorded):
  """The N > 1 control flow of bench.py end to end on a single-GPU box: two ranks on cuda:0 over gloo
  (RCCL refuses duplicate devices): shared scratch directory, rank-0 checkpoint + teacher hand-off,
  broadcast of the flat buffers, gradient all-reduce with the 1/N folded into Adam, max-over-ranks timing."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, PF_DIST_BACKEND='gloo', PF_SINGLE_DEVICE='1', TMPDIR=str(tmp_path))
  if recorded:
    env['PF_STEP_GRAPH_DIST'] = '1'                          # (round 6: N > 1 runs launch by launch unless this opts in)
  else:
    env.pop('PF_STEP_GRAPH_DIST', None)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
         '127.0.0.1', '--master-port', '29533' if recorded else '29535', os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4',
         '--event_steps', '2', '--warmup', '1', '--batch', '8', '--image_size', '64', '--no_cpu_baseline']
  out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
  rec = json.loads(line)
  assert rec['n_gpus'] == 2 and rec['config']['global_batch'] == 16 and rec['value'] > 0
  mg = rec['multi_gpu']
  assert mg['backend'] == 'gloo' and mg['params_identical_across_ranks'] is True, mg
  assert mg['buckets_launched_inside_backward'] > 0, mg      # the in-backward launches are armed by optimizer.backward()
  if not recorded:
    # the default for N > 1: launch-by-launch steps, the bucket all-reduces inside the backward pass
    assert mg['recorded_step'] is None and rec['config']['step_graph'] is None, (mg, rec['config'])
    return
  # PF_STEP_GRAPH_DIST=1: the step is recorded as two graphs around the exchange calls and replayed; the job STAYS on the
  # recorded step only if its replays are not slower than launch-by-launch steps (they are, with two ranks on one GPU over gloo)
  rs = mg['recorded_step']
  assert rs is not None and rs['graphs'] == 2 and rs['exchange_calls_between_graphs'] == 1 and rs['replayed_steps'] >= 2, mg
  assert rs['kept'] == (rs['replay_ms_per_step'] <= 1.1 * rs['launch_by_launch_ms_per_step']), rs
  print('   two ranks on one GPU: %d graphs per step, %d replays, %.1f ms per replayed step vs %.1f launch by launch -> %s' % (
      rs['graphs'], rs['replayed_steps'], rs['replay_ms_per_step'], rs['launch_by_launch_ms_per_step'],
      'kept' if rs['kept'] else 'back to launch-by-launch steps'))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: the RCCL path of utils/multi_gpu_wrapper.py')
def test_bench_two_gpus_over_rccl(tmp_path):
  """bench.py --gpus 2 on the nccl (= RCCL) backend, one rank per GPU (VERDICT r2 "next" 8): after the broadcast and three
  all-reduced Adam steps both ranks hold bit-identical parameters, and the bucketed all-reduce was launched from inside
  backward.  Skipped on the 1-GPU boxes of the build loop; the driver's multi-GPU box runs it."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, TMPDIR=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY='0', PF_ALLREDUCE_BUCKET=str(4 << 20))
  env.pop('PF_DIST_BACKEND', None)
  env.pop('PF_SINGLE_DEVICE', None)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
         '127.0.0.1', '--master-port', '29541', os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3',
         '--warmup', '1', '--batch', '16', '--image_size', '64', '--no_cpu_baseline']
  out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-3000:]
  rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
  mg = rec['multi_gpu']
  print('2 x MI355X over %s: %.0f images/s, %d of %d buckets launched inside backward, %.1f MB all-reduced per step' % (
      mg['backend'], rec['value'], mg['buckets_launched_inside_backward'], mg['buckets'], mg['allreduce_bytes_per_step'] / 1e6))
  assert rec['n_gpus'] == 2 and mg['backend'] == 'nccl'
  assert mg['params_identical_across_ranks'] is True
  assert mg['buckets'] >= 2 and mg['buckets_launched_inside_backward'] > 0


# =================================================================================================
# the recorded step (pocketflow_amd/step_graph.py): a hipGraph replay must BE the launch-by-launch step
# =================================================================================================

@pytest.mark.parametrize('case', ['uq_resnet50', 'ws_resnet20', 'cp_mobilenet'])
def test_step_graph_is_the_eager_step(tmp_path, case):
  """tests/step_graph_worker.py in a process of its own (a crash inside the HIP runtime's graph instantiation costs this test, not the
  suite; round 4: hipStreamEndCapture died whenever the previous step's autograd graph was still alive during the recording -- the
  worker keeps each step's result bound while it calls the next step, as the learners' train() loops do).  The three benchmarked learner kinds: ResNet-50 UQ w8/a8 +
  distillation (teacher forked inside the graph, Adam's alpha_t from device memory, graph suspended for two steps), ResNet-20
  weight sparsification (no teacher, Momentum, masks inside the recorded optimiser launch), MobileNet-v1 channel-pruned fine-tune
  (dropout mask through graph.step_feeders).  Recorded run vs launch-by-launch run of a second learner from the same checkpoint:
  same batches in the same order; losses within 1e-4, parameters within 2e-3 (bit-identical wherever every launch is ours)."""
  import json
  import subprocess
  import sys
  worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'step_graph_worker.py')
  r = subprocess.run([sys.executable, worker, case, str(tmp_path)], capture_output=True, text=True, timeout=900)
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith('STEP_GRAPH_RESULT ')]
  assert r.returncode == 0 and lines, 'worker rc %d\n%s\n%s' % (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
  res = json.loads(lines[-1][len('STEP_GRAPH_RESULT '):])
  print('   step graph %s: bit-identical %s, max parameter difference %.3e, last losses eager %s | graph %s' % (
      case, res['exact'], res['max_parameter_difference'], res['losses_eager'][-2:], res['losses_graph'][-2:]))


def test_backward_filter_side_queue_is_the_one_queue_step(tmp_path):
  """graph.WrwSide (round 6): the backward-filter launches of a pass on a second HIP stream, one join before the optimiser.  The same
  worker as above for ResNet-50 UQ + distillation: the RECORDED run forks them (edges of the hipGraph), the launch-by-launch run of
  the second learner keeps them in the one queue (PF_W_ONE_QUEUE_EAGER=1).  Deterministic kernels, same batches: losses, parameters,
  Adam slots and moving statistics must be bit-identical -- a filter gradient read before its launch finished, or a split workspace
  shared between the two queues, would show here."""
  import json
  import subprocess
  import sys
  worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'step_graph_worker.py')
  env = dict(os.environ, PF_W_ONE_QUEUE_EAGER='1')
  r = subprocess.run([sys.executable, worker, 'uq_resnet50', str(tmp_path)], capture_output=True, text=True, timeout=900, env=env)
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith('STEP_GRAPH_RESULT ')]
  assert r.returncode == 0 and lines, 'worker rc %d\n%s\n%s' % (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
  res = json.loads(lines[-1][len('STEP_GRAPH_RESULT '):])
  assert res['exact'] is True and res['max_parameter_difference'] == 0.0, res


def test_step_graph_with_two_ranks_is_two_graphs_around_the_exchange(tmp_path):
  """--enbl_step_graph with --enbl_multi_gpu (VERDICT r4 "next" 4; reference: Horovod's all-reduce is part of the compiled train
  graph, utils/multi_gpu_wrapper.py:83-98, learners/uniform_quantization/learner.py:246).  Two ranks share cuda:0 over gloo (the
  hook of test_bench_two_ranks_share_one_gpu); the worker's case holds that the recorded step -- two hipGraphs around the gradient
  exchange -- is bit for bit the launch-by-launch step on each rank, at a size where every launch of the step is this library's."""
  import json
  import subprocess
  import sys
  worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'step_graph_worker.py')
  env = dict(os.environ, PF_DIST_BACKEND='gloo', PF_SINGLE_DEVICE='1', TMPDIR=str(tmp_path), PF_STEP_GRAPH_DIST='1')   # (opt-in for library users since round 6)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', '29547', worker, 'uq_resnet50_two_ranks', str(tmp_path)]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith('STEP_GRAPH_RESULT ')]
  assert r.returncode == 0 and lines, 'worker rc %d\n%s\n%s' % (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
  res = json.loads(lines[-1][len('STEP_GRAPH_RESULT '):])
  print('   two ranks: %d graphs around %d exchange calls, bit-identical %s, last losses of the ranks %s' % (
      res['graphs'], res['actions'], res['exact'], res['last_loss_of_each_rank']))
  assert res['exact'] and res['graphs'] == 2
