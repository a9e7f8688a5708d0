"""Abstract learner -- same public surface as the reference's AbstractLearner
(learners/abstract_learner.py:41-158): __init__(sm_writer, model_helper), train(), evaluate(),
download_model(), auto_barrier(), is_primary_worker(), properties vars / trainable_vars / update_ops.

A learner owns one `Graph` per model scope instead of two tf.Graph + tf.Session pairs: the train and
eval "graphs" of the reference share nothing but variable VALUES (moved through checkpoints), so a
single variable store serves both, and `evaluate()` still round-trips through a checkpoint file
exactly like the reference does (uq learner.py:145-147,160).
"""
from __future__ import annotations

import os
import shutil
import subprocess
from abc import ABC, abstractmethod

import torch

from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.graph import Graph, to_device_images
from pocketflow_amd.utils import checkpoint
from pocketflow_amd.utils.misc_utils import MpiCommShim
from pocketflow_amd.utils.misc_utils import auto_barrier as auto_barrier_impl
from pocketflow_amd.utils.misc_utils import is_primary_worker as is_primary_worker_impl
from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw

flags.DEFINE_string('model_http_url', None, 'HTTP/HTTPS url for remote model files')
flags.DEFINE_integer('summ_step', 100, 'summarizaton step size')
flags.DEFINE_integer('save_step', 10000, 'model saving step size')
flags.DEFINE_string('save_path', './models/model.ckpt', 'model\'s save path')
flags.DEFINE_string('save_path_eval', './models_eval/model.ckpt', 'model\'s save path for evaluation')
flags.DEFINE_boolean('enbl_dst', False, 'enable the distillation loss for training')
flags.DEFINE_boolean('enbl_warm_start', False, 'enable warm start for training')
# --- flags with no reference counterpart (the reference is fp32 TF on whatever device TF picks) ---
flags.DEFINE_string('compute_dtype', 'float32', 'activation / matmul dtype: float32 (parity) | bfloat16 (MFMA)')
flags.DEFINE_integer('init_seed', 42, 'seed of the variable initialisers')
flags.DEFINE_string('ckpt_format', 'npz', "checkpoint files written by the learners: 'npz' | 'tf' (TensorFlow Saver-V2 "
                    "bundle, readable by the reference's tools); both are read transparently")
flags.DEFINE_boolean('fuse_conv1x1', True, 'bf16 mode: apply BN/ReLU/fake-quant inside the consuming 1x1 convolutions '
                     '(pf_conv.hip) instead of writing the activated tensor to HBM')
flags.DEFINE_boolean('enbl_step_graph', False, 'record the steady-state fine-tune step in a hipGraph after three eager steps and replay '
                     'it (single process; pocketflow_amd/step_graph.py).  Same results, one host call per step')
flags.DEFINE_integer('nb_iters_override', 0, 'if > 0, train() stops after this many iterations')
flags.DEFINE_integer('nb_eval_batches_override', 0, 'if > 0, evaluate() uses this many batches')


def compute_dtype():
  if FLAGS.compute_dtype in ('float32', 'fp32'):
    return torch.float32
  if FLAGS.compute_dtype in ('bfloat16', 'bf16'):
    return torch.bfloat16
  raise ValueError('unsupported --compute_dtype ' + str(FLAGS.compute_dtype))


def require_gpu() -> torch.device:
  """The hot path runs on an MI355X only -- no CPU fallback (the HIP library has none either)."""
  if not torch.cuda.is_available():
    raise RuntimeError('pocketflow_amd learners need a ROCm GPU (torch.cuda.is_available() is False); '
                       'the CPU oracle under oracle/ is test infrastructure, not a fallback')
  idx = mgw.device_index() if FLAGS.enbl_multi_gpu else 0
  torch.cuda.set_device(idx)
  return torch.device('cuda', idx)


def input_spec(model_helper):
  """A `meta` tensor with the shape of one training batch [B, H, W, C] (build-mode input)."""
  ds = getattr(model_helper, 'dataset_train', None)
  if ds is not None and hasattr(ds, 'image_shape'):
    return torch.empty((ds.batch_size,) + tuple(ds.image_shape), device='meta')
  images, __ = model_helper.build_dataset_train().get_next()
  return torch.empty(tuple(images.shape), device='meta')


def _detached(out):
  """The step's return value without its autograd graph.  The backward pass has run; nothing reads the graph afterwards -- but a
  caller that keeps the previous step's losses while it calls the next step (`log_rslt = self.train_step()` in every train() loop)
  would keep that step's whole graph alive, and a graph that is alive while the next step is RECORDED into a hipGraph crashed
  hipStreamEndCapture (round 4, found by bisecting tests/step_graph_worker.py: PF_W_DROP_OUT)."""
  if torch.is_tensor(out):
    return out.detach()
  if isinstance(out, dict):
    return {k: _detached(v) for k, v in out.items()}
  if isinstance(out, (tuple, list)):
    return type(out)(_detached(v) for v in out)
  return out


class AbstractLearner(ABC):  # pylint: disable=too-many-instance-attributes
  """Abstract class for learners: takes a ModelHelper (data pipeline + network definition) and
  either trains (periodically saving checkpoints) or restores and evaluates a model."""

  def __init__(self, sm_writer, model_helper):
    self.sm_writer = sm_writer
    self.data_scope = 'data'
    self.model_scope = 'model'

    if FLAGS.enbl_multi_gpu:
      mgw.init()
      self.mpi_comm = MpiCommShim()
    else:
      self.mpi_comm = None

    self.model_helper = model_helper
    self.build_dataset_train = model_helper.build_dataset_train
    self.build_dataset_eval = model_helper.build_dataset_eval
    self.forward_train = model_helper.forward_train
    self.forward_eval = model_helper.forward_eval
    self.calc_loss = model_helper.calc_loss
    self.setup_lrn_rate = model_helper.setup_lrn_rate
    self.warm_start = model_helper.warm_start
    self.dump_n_eval = model_helper.dump_n_eval
    self.model_name = model_helper.model_name
    self.dataset_name = model_helper.dataset_name
    self.forward_w_labels = model_helper.forward_w_labels

    self.ckpt_file = 'models_%s_at_%s.tar.gz' % (self.model_name, self.dataset_name)
    self.device = require_gpu()
    self.graph = None

  # ---------------------------------------------------------------------------------------------
  @abstractmethod
  def train(self):
    """Train a model and periodically produce checkpoint files."""

  @abstractmethod
  def evaluate(self):
    """Restore a model from the latest checkpoint files and then evaluate it."""

  def download_model(self):
    """Download remote model files and then uncompress (no-op when local files exist)."""
    if checkpoint.latest_checkpoint(os.path.dirname(FLAGS.save_path)) is not None:
      return
    if FLAGS.model_http_url is None:
      raise ValueError('local model files do not exist and <model_http_url> is not set')
    subprocess.call(['wget', os.path.join(FLAGS.model_http_url, self.ckpt_file)])
    if os.path.exists(self.ckpt_file):
      if os.path.isdir(os.path.dirname(FLAGS.save_path)):
        shutil.rmtree(os.path.dirname(FLAGS.save_path))
      subprocess.call(['tar', '-xvf', self.ckpt_file])
    else:
      raise FileNotFoundError(
          'pre-trained model not avaialable: {} / {}'.format(self.model_name, self.dataset_name))

  def auto_barrier(self):
    auto_barrier_impl(self.mpi_comm)

  @classmethod
  def is_primary_worker(cls, scope='global'):
    return is_primary_worker_impl(scope)

  @property
  def vars(self):
    """List of all global variables of the model scope."""
    return list(self.graph.store.vars)

  @property
  def trainable_vars(self):
    """List of all trainable variables of the model scope."""
    return self.graph.store.trainable_vars

  @property
  def update_ops(self):
    """BN moving-average updates are fused into the BN kernels; nothing to run separately."""
    return []

  # ---------------------------------------------------------------------------------------------
  # shared machinery (no reference counterpart: replaces tf.Graph / tf.Session plumbing)
  # ---------------------------------------------------------------------------------------------
  def build_graph(self, scope, separate_compute=False, requires_grad=True, before_finalize=None):
    """Declare the model's variables & ops on a new Graph by running forward_train in build mode.
    `before_finalize(graph)` may declare further variables (e.g. NUQ codebooks)."""
    graph = Graph(scope, self.device, compute_dtype())
    graph.fuse_conv1x1 = bool(FLAGS.fuse_conv1x1)
    spec = input_spec(self.model_helper)
    with graph.as_default():
      self.forward_train(spec)
    if before_finalize is not None:
      before_finalize(graph)
    graph.finalize(separate_compute=separate_compute, seed=FLAGS.init_seed, requires_grad=requires_grad)
    return graph

  def train_step(self, *args, **kwargs):
    """One fine-tune iteration (`sess.run(train_op)` of the reference learners): the learner's `_train_step_eager`, or -- with
    --enbl_step_graph -- the same step replayed from a hipGraph (step_graph.py; with --enbl_multi_gpu two graphs around the
    gradient-exchange calls; OPT-IN there, PF_STEP_GRAPH_DIST=1, until the two-graph chain has run over RCCL on two or more GPUs --
    the builder's boxes have one; bench.py opts in by itself and keeps the recorded step only where it is not slower)."""
    if args or kwargs or not FLAGS.enbl_step_graph or (FLAGS.enbl_multi_gpu and os.environ.get('PF_STEP_GRAPH_DIST', '0') == '0'):
      sg = getattr(self, '_step_graph', None)
      if sg is not None:
        sg.yield_to_eager()                                # the batches a ready step graph holds are the next ones in data order
      return _detached(self._train_step_eager(*args, **kwargs))
    from pocketflow_amd import step_graph
    return _detached(step_graph.of(self).step())

  def to_device(self, images, labels):
    x = to_device_images(images, self.graph)
    y = labels.to(self.device, non_blocking=True) if labels is not None else None
    return x, y

  def save_vars(self, save_path, global_step=None):
    return checkpoint.save(self.graph.store.export_numpy(), save_path, global_step, fmt=FLAGS.ckpt_format)

  def restore_vars(self, prefix, store=None, rename_scope=None, strict=True):
    (store or self.graph.store).load_numpy(checkpoint.load(prefix), strict=strict, rename_scope=rename_scope)
