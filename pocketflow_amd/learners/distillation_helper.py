"""Helper for training with the distillation loss (reference learners/distillation_helper.py:33-158).

A frozen full-precision teacher (scope `distilled_model`, weights copied from ./models with the first
path component renamed, :122-145) produces `logits_dst`; `calc_loss` is the temperature-softened
soft-label cross-entropy times `loss_w_dst` (:86-103) -- computed, with its gradient, by ONE fused HIP
kernel (pocketflow_amd.losses.distillation_loss -> pf_ce_distill_fwd_bwd).
"""
from __future__ import annotations

import logging
import os
import shutil

import torch

from pocketflow_amd import losses
from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.utils import checkpoint
from pocketflow_amd.utils.misc_utils import is_primary_worker

flags.DEFINE_float('loss_w_dst', 4.0, 'distillation loss\'s multiplier')
flags.DEFINE_float('tempr_dst', 4.0, 'temperature in the distillation loss')
flags.DEFINE_string('save_path_dst', './models_dst/model.ckpt', 'distillation model\'s save path')
flags.DEFINE_boolean('dst_eval_teacher', True, 'evaluate the teacher once at construction (as the reference does)')

log = logging.getLogger('pocketflow_amd')


class DistillationHelper(object):
  """Other learners call calc_logits() for the teacher's logits and calc_loss() for the loss."""

  def __init__(self, sm_writer, model_helper, mpi_comm):
    self.model_scope = 'distilled_model'  # to distinguish from models created by other learners
    from pocketflow_amd.learners.full_precision.learner import FullPrecLearner
    self.learner = FullPrecLearner(sm_writer, model_helper, self.model_scope, enbl_dst=False)
    self.model_helper = model_helper

    if is_primary_worker('local'):
      self.__initialize()
    if FLAGS.enbl_multi_gpu:
      mpi_comm.Barrier()
    if not is_primary_worker('local'):
      self.__restore(rename_only=False)

  def calc_logits(self, sess, images):
    """Teacher logits for `images` (already a device compute tensor); never carries gradients."""
    g = self.learner.graph
    with torch.no_grad():
      with g.as_default():
        logits = self.learner.forward_eval(images)
    return logits.detach()

  @classmethod
  def calc_loss(cls, logits_pri, logits_dst):
    """loss_w_dst * softmax_cross_entropy(softmax(logits_dst / T), logits_pri / T)."""
    return losses.distillation_loss(logits_pri, logits_dst, FLAGS.tempr_dst, FLAGS.loss_w_dst)

  def __initialize(self):
    """Copy ./models -> ./models_dst (after download_model), rename the scope, restore, evaluate."""
    self.learner.download_model()
    dst_dir = os.path.dirname(FLAGS.save_path_dst)
    if os.path.isdir(dst_dir):
      shutil.rmtree(dst_dir)
    shutil.copytree(os.path.dirname(FLAGS.save_path), dst_dir)
    self.__restore(rename_only=True)
    self.__restore(rename_only=False)
    if FLAGS.dst_eval_teacher:
      self.__evaluate()

  def __restore(self, rename_only):
    ckpt_dir = os.path.dirname(FLAGS.save_path_dst)
    prefix = checkpoint.latest_checkpoint(ckpt_dir)
    if rename_only:
      values = checkpoint.load(prefix)
      renamed = {}
      for name_old, val in values.items():
        name_new = self.model_scope + '/' + '/'.join(name_old.split('/')[1:])
        renamed[name_new] = val
      checkpoint.save(renamed, prefix)
      return
    self.learner.restore_vars(prefix)
    self.learner.graph.frozen = True
    log.info('model restored from ' + prefix)

  def __evaluate(self):
    rslt = self.learner.run_eval()
    log.info('teacher: %s', rslt)
