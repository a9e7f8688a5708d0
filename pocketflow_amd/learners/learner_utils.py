"""Learner factory (reference learners/learner_utils.py:33-66) + a helper that writes the
"pre-trained model" checkpoint the compression learners start from when no real one exists."""
from __future__ import annotations

import os

import torch

from pocketflow_amd.flags import FLAGS, flags

flags.DEFINE_string('learner', 'full-prec', 'learner\'s name')
flags.DEFINE_string('exec_mode', 'train', 'execution mode: train / eval')
flags.DEFINE_boolean('debug', False, 'debugging information')
flags.DEFINE_string('log_dir', './logs', 'logging directory')


def create_learner(sm_writer, model_helper):
  """Create the learner as specified by FLAGS.learner."""
  learner = None
  if FLAGS.learner == 'full-prec':
    from pocketflow_amd.learners.full_precision.learner import FullPrecLearner
    learner = FullPrecLearner(sm_writer, model_helper)
  elif FLAGS.learner == 'weight-sparse':
    from pocketflow_amd.learners.weight_sparsification.learner import WeightSparseLearner
    learner = WeightSparseLearner(sm_writer, model_helper)
  elif FLAGS.learner == 'channel':
    from pocketflow_amd.learners.channel_pruning.learner import ChannelPrunedLearner
    learner = ChannelPrunedLearner(sm_writer, model_helper)
  elif FLAGS.learner == 'uniform':
    from pocketflow_amd.learners.uniform_quantization.learner import UniformQuantLearner
    learner = UniformQuantLearner(sm_writer, model_helper)
  elif FLAGS.learner == 'non-uniform':
    from pocketflow_amd.learners.nonuniform_quantization.learner import NonUniformQuantLearner
    learner = NonUniformQuantLearner(sm_writer, model_helper)
  elif FLAGS.learner == 'chn-pruned-gpu':
    from pocketflow_amd.learners.channel_pruning_gpu.learner import ChannelPrunedGpuLearner
    learner = ChannelPrunedGpuLearner(sm_writer, model_helper)
  elif FLAGS.learner in ('chn-pruned-rmt', 'dis-chn-pruned', 'uniform-tf'):
    raise ValueError('learner %r is outside the MI355X hot path (SURVEY section 2, rows 9-12)' % FLAGS.learner)
  else:
    raise ValueError('unrecognized learner\'s name: ' + FLAGS.learner)
  return learner


def create_synthetic_checkpoint(model_helper, seed=None):
  """Write ./models/model.ckpt-0 with seeded initial weights: stands in for the pre-trained archive
  `models_<model>_at_<dataset>.tar.gz` (abstract_learner.py:105-125), which needs the network."""
  from pocketflow_amd.graph import Graph
  from pocketflow_amd.utils import checkpoint
  if checkpoint.latest_checkpoint(os.path.dirname(FLAGS.save_path)) is not None:
    return
  graph = Graph('model', 'cpu', torch.float32)
  from pocketflow_amd.learners.abstract_learner import input_spec
  with graph.as_default():
    model_helper.forward_train(input_spec(model_helper))
  graph.finalize(seed=FLAGS.init_seed if seed is None else seed, requires_grad=False)
  checkpoint.save(graph.store.export_numpy(), FLAGS.save_path, 0)
