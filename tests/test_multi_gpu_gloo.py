"""The N > 1 path on CPU: world_size 2, backend gloo (RCCL needs GPUs; the wrapper picks gloo when
torch.cuda is unavailable).  Covers the seven-method MultiGpuWrapper surface, the flat-buffer gradient
all-reduce with Horovod's averaging folded into g_scale, broadcast_global_variables, the mpi_comm
shim and the per-rank data seeds -- i.e. everything of SURVEY 8(e) except the device kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, out_dir):
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  import torch.distributed as dist
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd.graph import Graph
  from pocketflow_amd.nets.lenet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.abstract_learner import input_spec
  from pocketflow_amd.optim import DistributedFlatOptimizer
  from pocketflow_amd.utils.misc_utils import MpiCommShim, auto_barrier, is_primary_worker
  from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw
  FLAGS.enbl_multi_gpu = True
  FLAGS.batch_size = 8
  mgw.init()
  assert (mgw.size(), mgw.rank(), mgw.local_size(), mgw.local_rank()) == (world, rank, world, rank)
  assert is_primary_worker('global') == (rank == 0) and is_primary_worker('local') == (rank == 0)
  mh = ModelHelper()
  g = Graph('model', 'cpu', torch.float32)
  with g.as_default():
    mh.forward_train(input_spec(mh))
  g.finalize(seed=100 + rank, requires_grad=True)          # DIFFERENT initial weights per rank
  st = g.store

  class _Opt(object):                                      # the wrapped optimiser: record what apply gets
    store, g_scale, applied, weight_decay = st, None, None, 0.0
    w_grad_src = o_grad_src = None

    def state_tensors(self):
      return [self.slot]

    def apply_gradients(self, lr):
      self.applied = (lr, self.g_scale, self.w_grad_src.clone(), self.o_grad_src.clone(), self.weight_decay)
  opt = _Opt()
  opt.slot = torch.full((4,), float(rank))
  dopt = mgw.DistributedOptimizer(opt)
  assert isinstance(dopt, DistributedFlatOptimizer)
  # 1) broadcast_global_variables(0): parameters, BN state, optimiser slots all become rank 0's
  w_before = st.w_master.clone()
  mgw.broadcast_global_variables(0, [st], [dopt])()
  gathered = [torch.zeros_like(st.w_master) for _ in range(world)]
  dist.all_gather(gathered, st.w_master)
  assert all(torch.equal(gathered[0], t) for t in gathered)
  assert rank == 0 or not torch.equal(w_before, st.w_master)
  assert float(opt.slot[0]) == 0.0
  # 2) gradient exchange: sum over ranks in the flat buffers, average via g_scale = 1 / N
  st.w_grad.copy_(torch.arange(st.w_grad.numel(), dtype=torch.float32) * (rank + 1))
  st.o_grad.fill_(float(rank + 1))
  dopt.weight_decay = 5e-4                                 # learners set attributes on the WRAPPER
  assert opt.weight_decay == 5e-4 and dopt.weight_decay == 5e-4
  dopt.compute_gradients()
  dopt.apply_gradients(0.5)
  lr, g_scale, wg, og, wd = opt.applied
  tot = sum(r + 1 for r in range(world))
  assert lr == 0.5 and g_scale == 1.0 / world and wd == 5e-4
  assert torch.equal(wg, torch.arange(st.w_grad.numel(), dtype=torch.float32) * tot)
  assert torch.equal(og, torch.full_like(og, float(tot)))
  # 2b) the same exchange launched from INSIDE backward: producers report variables in reverse creation order,
  # buckets are reduced as they complete; a variable reporting twice falls back to a blocking re-reduction
  from pocketflow_amd.optim import GradReducer
  red = GradReducer(st, bucket_elems=1 << 12, reduce_dtype=torch.float32, overlap=True)
  dopt.reducer = red
  assert len(red.buckets) >= 2 and red.buckets[0][0] == 0 and red.buckets[-1][1] == st.w_size
  wvars = sorted([v for v in st.vars if v.group == 'W'], key=lambda v: -v.offset)
  for dirty in (False, True):
    # between two training steps rank 0 alone runs a backward pass (LayerwiseTuner under
    # *_enbl_rl_layerwise_tune, regression-gradient helpers) while the others wait in a barrier.  In-backward launching
    # is opt-in per step, so those reports must launch NOTHING -- a lone all-reduce would pair with the barrier -- and
    # must not leak seen / pending state into the next armed cycle.
    if rank == 0:
      for v in wvars:
        st.notify_grad(v)
      assert not any(red.launched) and not red.handles and not red.armed
    dist.barrier()
    st.w_grad.copy_(torch.arange(st.w_grad.numel(), dtype=torch.float32) * (rank + 2))
    st.o_grad.fill_(float(rank + 2))
    red.arm()                                              # what DistributedFlatOptimizer.backward() does around loss.backward()
    for v in wvars:
      st.notify_grad(v)
    red.disarm()
    if dirty:
      st.w_grad.mul_(2.0)                                  # a second backward pass accumulated into the buffer
      st.notify_grad(wvars[0])
    else:
      assert all(red.launched), 'every bucket should be in flight before compute_gradients()'
    dopt.compute_gradients()
    dopt.apply_gradients(0.25)
    _, g_scale, wg, og, _ = opt.applied
    tot2 = sum(r + 2 for r in range(world)) * (2 if dirty else 1)
    assert g_scale == 1.0 / world
    assert torch.equal(wg, torch.arange(st.w_grad.numel(), dtype=torch.float32) * tot2)
    assert red.n_overlapped == (0 if dirty else len(red.buckets))
  # 2c) bf16 gradient buffer, float32 reduction: the sum is exact in float32 (no bf16 rounding of partial sums)
  wg16 = (torch.arange(st.w_grad.numel(), dtype=torch.float32) % 251 + 0.5 * rank).to(torch.bfloat16)
  st_w_grad_fp32 = st.w_grad
  st.w_grad = wg16.clone()
  red16 = GradReducer(st, bucket_elems=1 << 12, reduce_dtype=torch.float32, overlap=False)
  dopt.reducer = red16
  dopt.compute_gradients()
  both = [torch.zeros_like(wg16) for _ in range(world)]
  dist.all_gather(both, wg16)
  assert opt.w_grad_src.dtype == torch.float32
  assert torch.equal(opt.w_grad_src, sum(t.float() for t in both))
  st.w_grad = st_w_grad_fp32
  # 3) mpi_comm shim: pickled-object broadcast (channel-pruning masks / decisions) + barrier
  comm = MpiCommShim()
  obj = comm.bcast({'conv1': [np.array([True, False]), np.array([True])]} if rank == 0 else None, root=0)
  assert obj['conv1'][0].tolist() == [True, False]
  auto_barrier(comm)
  # 4) per-rank data stream (reference: file-level shard(size, rank))
  it = mh.build_dataset_train()
  images, _ = it.get_next()
  np.save(os.path.join(out_dir, 'img_%d.npy' % rank), images.numpy()[:1])
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_gloo(tmp_path):
  port = _free_port()
  mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  a, b = np.load(tmp_path / 'img_0.npy'), np.load(tmp_path / 'img_1.npy')
  assert a.shape == b.shape and not np.array_equal(a, b)


def test_wrapper_without_launcher_raises_like_the_reference():
  """utils/multi_gpu_wrapper.py:41-44: NameError('module <mgw> not imported') when Horovod is absent."""
  from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw
  import torch.distributed as dist
  if dist.is_initialized() or 'RANK' in os.environ:
    pytest.skip('running under a launcher')
  mgw._initialized = False
  with pytest.raises(NameError, match='module <mgw> not imported'):
    mgw.init()
  assert mgw.size() == 1 and mgw.rank() == 0


def test_bf16_sum_of_8_ranks_is_why_the_reduction_is_float32():
  """a bf16 gradient buffer summed over 8 ranks in bf16 (what `--allreduce_dtype compute` does)
  carries 8 mantissa bits through 7 additions; the float32 staging path (default) is exact up to float32 rounding.
  Arithmetic emulation of a ring reduction (partial sums rounded to the wire dtype at every hop)."""
  rng = np.random.RandomState(0)
  g = torch.from_numpy((rng.randn(8, 1 << 16) * 1e-3).astype(np.float32)).to(torch.bfloat16)
  exact = g.double().sum(0)
  ring16 = g[0].clone()
  for r in range(1, 8):
    ring16 = (ring16 + g[r])                                # bf16 + bf16 -> bf16 (one rounding per hop)
  ring32 = g.float().sum(0)
  scale = float(exact.abs().mean())
  e16 = float((ring16.double() - exact).abs().max()) / scale
  e32 = float((ring32.double() - exact).abs().max()) / scale
  assert e32 < 1e-5, e32
  assert 1e-3 < e16 < 5e-2, e16                             # ~2^-8 per hop: above the 1e-3 parity bar of north_star


def _agree_worker(rank, world, port, fail_rank):
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  import contextlib
  import types
  import torch.distributed as dist
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from pocketflow_amd import step_graph as SG

  class Red(object):
    recorder = 'the backend of the recording'

    def _active(self):
      return True

  class Opt(object):
    reducer, hyper_external = Red(), False

  class Gr(object):
    capturing = False

  class Lrn(object):
    optimizer, graph, device = Opt(), Gr(), 'cpu'

    def _train_step_eager(self):
      return 'eager'

  class Be(object):
    def warm(self):
      return contextlib.nullcontext()

    def recover(self):
      pass
  sg = SG.StepGraph(Lrn(), Be())

  def rec(self):
    if rank == fail_rank:
      raise RuntimeError('this rank cannot capture')
    self.state = 'ready'
  sg._record = types.MethodType(rec, sg)
  sg._replay = lambda: 'replay'
  outs = [sg.step() for _ in range(SG.StepGraph.WARM + 2)]
  if fail_rank is None:
    assert outs == ['eager'] * SG.StepGraph.WARM + ['replay'] * 2 and sg.state == 'ready', (rank, outs, sg.state)
  else:
    # EVERY rank stays launch by launch -- also the one whose recording succeeded -- and the reducer forgets the backend
    assert outs == ['eager'] * (SG.StepGraph.WARM + 2) and sg.state == 'failed', (rank, outs, sg.state)
    assert Lrn.optimizer.reducer.recorder is None
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('fail_rank', [1, None])
def test_recorded_step_is_kept_only_if_every_rank_recorded_it(fail_rank):
  """with --enbl_step_graph a multi-rank job decides TOGETHER whether it replays (MIN all-reduce of the success
  flag right after the recording attempt, step_graph.StepGraph._ranks_agree): a rank that fell back alone would issue its bucket
  all-reduces in another order and size than the ranks that replay."""
  mp.spawn(_agree_worker, args=(2, _free_port(), fail_rank), nprocs=2, join=True)
