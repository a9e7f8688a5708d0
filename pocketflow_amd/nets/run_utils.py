"""Shared `main()` of the nets/*_run.py entry scripts (reference nets/resnet_at_ilsvrc12_run.py:33-69):
parse flags (unknown flags are fatal), build the ModelHelper and the learner named by --learner, then
train or download+evaluate; a ValueError exits with status 1."""
from __future__ import annotations

import logging
import traceback

from pocketflow_amd.flags import FLAGS


class SummaryWriter(object):
  """The two tf.summary.FileWriter methods the learners call, as JSON lines under --log_dir."""

  def __init__(self, log_dir):
    import os
    os.makedirs(log_dir, exist_ok=True)
    self.path = os.path.join(log_dir, 'summaries.jsonl')

  def add_graph(self, graph):
    pass

  def add_summary(self, summary, global_step=None):
    import json
    with open(self.path, 'a') as f:
      f.write(json.dumps({'step': global_step, **{k: float(v) for k, v in dict(summary).items()}}) + '\n')


def run_main(model_helper_cls, argv=None) -> int:
  # flags are defined at import time by the modules that need them (as in the reference)
  import pocketflow_amd.learners.learner_utils as learner_utils
  import pocketflow_amd.learners.abstract_learner  # noqa: F401
  import pocketflow_amd.learners.distillation_helper  # noqa: F401
  import pocketflow_amd.learners.full_precision.learner  # noqa: F401
  import pocketflow_amd.learners.uniform_quantization.learner  # noqa: F401
  import pocketflow_amd.learners.nonuniform_quantization.learner  # noqa: F401
  import pocketflow_amd.learners.weight_sparsification.learner  # noqa: F401
  import pocketflow_amd.learners.channel_pruning.learner  # noqa: F401
  import pocketflow_amd.learners.channel_pruning_gpu.learner  # noqa: F401
  FLAGS.parse(argv)
  try:
    logging.basicConfig(level=logging.DEBUG if FLAGS.debug else logging.INFO,
                        format='%(levelname)s:%(name)s:%(message)s')
    log = logging.getLogger('pocketflow_amd')
    sm_writer = SummaryWriter(FLAGS.log_dir)
    log.info('FLAGS:')
    for key, value in FLAGS.flag_values_dict().items():
      log.info('{}: {}'.format(key, value))
    model_helper = model_helper_cls()
    learner = learner_utils.create_learner(sm_writer, model_helper)
    if FLAGS.exec_mode == 'train':
      learner.train()
    elif FLAGS.exec_mode == 'eval':
      learner.download_model()
      learner.evaluate()
    else:
      raise ValueError('unrecognized execution mode: ' + FLAGS.exec_mode)
    return 0
  except ValueError:
    traceback.print_exc()
    return 1
