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
    store, g_scale, applied = st, None, None

    def state_tensors(self):
      return [self.slot]

    def apply_gradients(self, lr):
      self.applied = (lr, self.g_scale, st.w_grad.clone(), st.o_grad.clone())
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
  dopt.compute_gradients()
  dopt.apply_gradients(0.5)
  lr, g_scale, wg, og = opt.applied
  tot = sum(r + 1 for r in range(world))
  assert lr == 0.5 and g_scale == 1.0 / world
  assert torch.equal(wg, torch.arange(st.w_grad.numel(), dtype=torch.float32) * tot)
  assert torch.equal(og, torch.full_like(og, float(tot)))
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
