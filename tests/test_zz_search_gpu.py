"""Hyper-parameter searches on the GPU: the roll-outs re-run the hot path through the HIP
library -- per-layer bit widths fed to the segment kernels, mask kernels + inference-mode BN with gradients for
the pruning-ratio search, tapped forwards for the channel-pruning search.  Small configurations: the CPU tests
(tests/test_learners_cpu.py, tests/test_rl_golden.py, tests/test_ddpg_agent.py) carry the parity checks; these
check that the same host code drives the real kernels.  (File name sorts last on purpose.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(tmp_path, **kw):
  from pocketflow_amd.flags import FLAGS
  import pocketflow_amd.learners.learner_utils  # noqa: F401  (defines flags)
  import pocketflow_amd.learners.abstract_learner  # noqa: F401
  import pocketflow_amd.rl_agents.ddpg.agent  # noqa: F401
  import pocketflow_amd.learners.weight_sparsification.learner  # noqa: F401
  import pocketflow_amd.learners.channel_pruning.learner  # noqa: F401
  import pocketflow_amd.learners.uniform_quantization.learner  # noqa: F401
  import pocketflow_amd.nets.resnet_at_cifar10  # noqa: F401
  import pocketflow_amd.nets.mobilenet_at_ilsvrc12  # noqa: F401
  import pocketflow_amd.learners.channel_pruning_gpu.learner  # noqa: F401
  FLAGS.save_path = str(tmp_path / 'models' / 'model.ckpt')
  FLAGS.save_path_eval = str(tmp_path / 'models_eval' / 'model.ckpt')
  FLAGS.synthetic_pool = 2
  FLAGS.compute_dtype = 'float32'
  FLAGS.nb_eval_batches_override = 2
  for k, v in kw.items():
    setattr(FLAGS, k, v)
  return FLAGS


def test_inference_mode_bn_gradients_on_gpu():
  """graph._BnEvalAct: pf_bn_eval_scale_shift + pf_bn_act_quant_apply forward, pf_bn_bwd_stats / finalize +
  pf_bn_bwd_apply (zero sums) backward, against torch autograd; float32 and bfloat16 activations."""
  import pocketflow_amd.graph as G
  for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 3e-2)):
    g = G.Graph('model', 'cuda', dtype)
    bn = G.BatchNormAct(g, 'bn', 64, 'Relu', 0.997, 1e-5)
    g.finalize(requires_grad=True)
    rng = np.random.RandomState(0)
    dev = torch.device('cuda')
    with torch.no_grad():
      bn.gamma.tensor.copy_(torch.from_numpy((1 + 0.2 * rng.randn(64)).astype(np.float32)))
      bn.beta.tensor.copy_(torch.from_numpy((0.2 * rng.randn(64)).astype(np.float32)))
      bn.moving_mean.tensor.copy_(torch.from_numpy((0.3 * rng.randn(64)).astype(np.float32)))
      bn.moving_var.tensor.copy_(torch.from_numpy((0.5 + rng.rand(64)).astype(np.float32)))
    x0 = torch.from_numpy(rng.randn(8, 64, 7, 7).astype(np.float32)).to(dev)
    up = torch.from_numpy(rng.randn(8, 64, 7, 7).astype(np.float32)).to(dev)
    x = x0.to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g.training = False
    with g.as_default():
      y = bn(x)
    (y.float() * up).sum().backward()
    mm, mv = bn.moving_mean.tensor.clone(), bn.moving_var.tensor.clone()
    xr = x.detach().float().clone().requires_grad_(True)
    gam = bn.gamma.tensor.detach().clone().requires_grad_(True)
    bet = bn.beta.tensor.detach().clone().requires_grad_(True)
    v = lambda t: t.view(1, -1, 1, 1)
    yr = torch.relu((xr - v(mm)) * torch.rsqrt(v(mv) + 1e-5) * v(gam) + v(bet))
    (yr * up).sum().backward()
    scale = lambda t: float(t.detach().abs().max())
    assert float((y.detach().float() - yr.detach()).abs().max()) <= tol * max(1.0, scale(yr))
    assert float((x.grad.float() - xr.grad).abs().max()) <= tol * max(1.0, scale(xr.grad))
    assert float((bn.gamma.tensor.grad - gam.grad).abs().max()) <= tol * max(1.0, scale(gam.grad))
    assert float((bn.beta.tensor.grad - bet.grad).abs().max()) <= tol * max(1.0, scale(bet.grad))
    assert torch.equal(bn.moving_mean.tensor, mm) and torch.equal(bn.moving_var.tensor, mv)


def test_uq_bit_width_search(tmp_path):
  from pocketflow_amd.nets.lenet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.uniform_quantization.learner import UniformQuantLearner
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  FLAGS = _setup(tmp_path, batch_size=16, batch_size_eval=16, uql_weight_bits=4, uql_activation_bits=32, uql_quantize_all_layers=True,
                 uql_enbl_rl_agent=True, uql_nb_rlouts=4, uql_tune_global_steps=2, uql_equivalent_bits=5,
                 uql_tune_save_path=str(tmp_path / 'rl_tune' / 'model.ckpt'),
                 uql_save_quant_model_path=str(tmp_path / 'uql' / 'm.ckpt'), ddpg_seed=7, nb_iters_override=2)
  mh = ModelHelper()
  create_synthetic_checkpoint(mh)
  lrn = UniformQuantLearner(None, mh)
  w_bits = lrn.optimal_w_bit_list
  assert len(w_bits) == lrn.statistics['nb_matmuls'] == 4 and lrn.optimal_a_bit_list == [32] * 3
  assert all(2 <= b <= 8 for b in w_bits)
  assert sum(b * k for b, k in zip(w_bits, lrn.statistics['num_weights'])) <= 5 * sum(lrn.statistics['num_weights'])
  # the searched widths reach the segment table the kernels read
  segs = np.frombuffer(lrn.uni_quant.plan.segs.cpu().numpy().tobytes(), dtype=lrn.uni_quant.plan.segs_host.dtype)
  assert [int(b) for b in segs['bits'] if b > 0] == w_bits
  rslt = lrn.train()
  assert np.isfinite(rslt['loss'])


def test_ws_pruning_ratio_search(tmp_path):
  from pocketflow_amd.nets.resnet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  from pocketflow_amd.learners.weight_sparsification.pr_optimizer import PROptimizer
  FLAGS = _setup(tmp_path, batch_size=16, batch_size_eval=16, resnet_size=8, nb_classes=10, ws_prune_ratio=0.6,
                 ws_prune_ratio_prtl='optimal', ws_nb_rlouts=2, ws_nb_rlouts_min=1, ws_nb_iters_rg=2, ws_nb_iters_ft=3,
                 ws_nb_iters_feval=2, ws_lrn_rate_rg=1e-3, ddpg_seed=5)
  mh = ModelHelper()
  create_synthetic_checkpoint(mh)
  opt = PROptimizer(mh, None)
  pairs = opt.run()
  ratios = np.array([r for _, r in pairs])
  n = np.array([v.numel for v in opt.vars_full['maskable']], dtype=np.float64)
  assert len(pairs) == 11 and ratios[0] == 0.0 and ratios[-1] == 0.0
  assert np.sum(n * ratios) / np.sum(n) >= 0.6 - 1e-4
  assert len(opt.reward_history) == 2 and all(np.isfinite(opt.reward_history))
  st = opt.graph_prnd.store
  for v in opt.vars_prnd['maskable']:
    sl = slice(v.offset, v.offset + v.numel)
    assert float((st.w_master[sl] * (1 - opt.masks[sl])).abs().max()) == 0.0       # masked through regression + fine-tune
    assert torch.isfinite(st.w_master[sl]).all()
  # masks of the last roll-out: exact sparsity of the magnitude threshold
  last = [float('%f' % r) for r in opt._PROptimizer__bcast([r for _, r in pairs])]
  assert len(last) == 11


def test_channel_pruning_search(tmp_path):
  from pocketflow_amd.nets.mobilenet_at_ilsvrc12 import ModelHelper
  from pocketflow_amd.learners.channel_pruning.learner import ChannelPrunedLearner
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  FLAGS = _setup(tmp_path, batch_size=8, batch_size_eval=8, image_size=32, nb_classes=17, mobilenet_depth_mult=0.25,
                 cp_prune_option='auto', cp_preserve_ratio=0.5, cp_nb_batches=4, cp_nb_points_per_layer=10,
                 cp_nb_rlouts=2, cp_nb_rlouts_min=1,
                 cp_channel_pruned_path=str(tmp_path / 'models' / 'pruned_model.ckpt'),
                 cp_best_path=str(tmp_path / 'models' / 'best_model.ckpt'),
                 cp_original_path=str(tmp_path / 'models' / 'original_model.ckpt'),
                 nb_iters_override=2, summ_step=2, synthetic_pool=4, ddpg_seed=3)
  mh = ModelHelper()
  create_synthetic_checkpoint(mh)
  lrn = ChannelPrunedLearner(None, mh)
  rslt = lrn.train()
  assert np.isfinite(rslt['loss'])
  strategy, acc, flops = lrn.bestinfo
  assert len(lrn.reward_history) == 2 and len(strategy) == len(lrn.pruner.thisconvs)
  assert strategy[0] == 1.0 and strategy[-1] == 1 and all(0 < r <= 1 for r in strategy)
  assert lrn.pruner.preserve_ratio <= 0.5 + 0.08


def test_channel_pruned_gpu_learner(tmp_path):
  """'chn-pruned-gpu': proximal-gradient channel selection against the full network, all on the device."""
  from pocketflow_amd.nets.resnet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.learner_utils import create_learner, create_synthetic_checkpoint
  import pocketflow_amd.learners.channel_pruning_gpu.learner as CPG
  FLAGS = _setup(tmp_path, batch_size=16, batch_size_eval=16, resnet_size=8, nb_classes=10, learner='chn-pruned-gpu',
                 cpg_prune_ratio=0.5, cpg_nb_iters_layer=6, cpg_lrn_rate_pgd_init=1e-6,
                 cpg_save_path=str(tmp_path / 'cpg' / 'model.ckpt'), cpg_save_path_eval=str(tmp_path / 'cpg_eval' / 'model.ckpt'),
                 nb_iters_override=3, summ_step=2)
  mh = ModelHelper()
  create_synthetic_checkpoint(mh)
  lrn = create_learner(None, mh)
  assert isinstance(lrn, CPG.ChannelPrunedGpuLearner)
  rslt = lrn.train()
  assert np.isfinite(rslt['loss']) and 0.0 < rslt['pr_msk'] < 0.6
  vals = lrn.graph.store.export_numpy()
  for idx, var in enumerate(lrn.vars_prnd['maskable']):
    dead = np.all(vals[var.name] == 0, axis=(0, 1, 3))
    if idx in (0, lrn.nb_layers - 1):
      assert not dead.any()
    else:
      assert vals[var.name].shape[2] // 2 <= dead.sum() < vals[var.name].shape[2]
