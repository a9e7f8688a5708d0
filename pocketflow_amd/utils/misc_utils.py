"""auto_barrier / is_primary_worker (reference utils/misc_utils.py:25-52) over torch.distributed."""
from __future__ import annotations

import torch.distributed as dist

from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw

flags.DEFINE_boolean('enbl_multi_gpu', False, 'enable multi-GPU training')
flags.DEFINE_string('allreduce_dtype', 'float32', "dtype the gradient all-reduce sums in: 'float32' (bf16 gradient "
                    "buffers are widened first) | 'compute' (the buffer's own dtype: half the bytes on xGMI)")
flags.DEFINE_float('dist_timeout_hours', 24.0, 'collective timeout of the process group: rank-0-only phases (channel '
                   'selection, searches, model download) keep the other ranks at a barrier for far longer than the '
                   'RCCL default of 10 minutes')


def auto_barrier(mpi_comm=None):
  """Insert a barrier for multi-GPU training, or pass for single-GPU training."""
  if FLAGS.enbl_multi_gpu and dist.is_initialized():
    dist.barrier()


def is_primary_worker(scope='global'):
  """Whether this is the primary worker of all nodes ('global') or of the current node ('local')."""
  if scope == 'global':
    return True if not FLAGS.enbl_multi_gpu else mgw.rank() == 0
  elif scope == 'local':
    return True if not FLAGS.enbl_multi_gpu else mgw.local_rank() == 0
  else:
    raise ValueError('unrecognized worker scope: ' + scope)


class MpiCommShim(object):
  """The two mpi4py calls the learners use (Barrier, bcast of picklable objects) over torch.distributed."""

  def Barrier(self):
    if dist.is_initialized():
      dist.barrier()

  def bcast(self, obj, root=0):
    if not dist.is_initialized() or dist.get_world_size() == 1:
      return obj
    box = [obj]
    dist.broadcast_object_list(box, src=root)
    return box[0]
