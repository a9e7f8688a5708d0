"""Full-precision learner (no compression): the Momentum-SGD loop every other learner is a variant
of, and the container of the frozen teacher for distillation.
Reference: learners/full_precision/learner.py:30-228.
"""
from __future__ import annotations

import logging
import os
from timeit import default_timer as timer

import numpy as np
import torch

from pocketflow_amd.flags import FLAGS
from pocketflow_amd.learners.abstract_learner import AbstractLearner
from pocketflow_amd.optim import FlatOptimizer
from pocketflow_amd.utils import checkpoint
from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw

log = logging.getLogger('pocketflow_amd')


class FullPrecLearner(AbstractLearner):  # pylint: disable=too-many-instance-attributes
  """Full-precision learner (no model compression applied)."""

  def __init__(self, sm_writer, model_helper, model_scope=None, enbl_dst=None):
    super(FullPrecLearner, self).__init__(sm_writer, model_helper)
    if model_scope is not None:
      self.model_scope = model_scope
    self.enbl_dst = enbl_dst if enbl_dst is not None else FLAGS.enbl_dst

    if self.enbl_dst:
      from pocketflow_amd.learners.distillation_helper import DistillationHelper
      self.helper_dst = DistillationHelper(sm_writer, model_helper, self.mpi_comm)
    self.__build()

  def __build(self):
    """Declare the model; optimiser, schedule and data iterators (train + eval "graphs")."""
    is_teacher = self.model_scope != 'model'
    self.graph = self.build_graph(self.model_scope, requires_grad=not is_teacher)
    if is_teacher:
      self.graph.frozen = True
    self.iter_train = self.build_dataset_train().to(self.device)
    self.iter_eval = self.build_dataset_eval().to(self.device)
    self.global_step = 0
    self.lrn_rate, self.nb_iters_train = self.setup_lrn_rate(self.global_step)
    if not is_teacher:
      optimizer = FlatOptimizer(self.graph.store, 'momentum', momentum=FLAGS.momentum)
      if FLAGS.enbl_multi_gpu:
        optimizer = mgw.DistributedOptimizer(optimizer)
        self.bcast_op = mgw.broadcast_global_variables(0, [self.graph.store], [optimizer])
      self.optimizer = optimizer
    self.log_op_names = ['lr', 'loss']
    self.eval_op_names = ['loss']

  # ---------------------------------------------------------------------------------------------
  def train_step(self):
    """One `sess.run(train_op)`: data -> [teacher fwd] -> fwd -> loss -> bwd -> [all-reduce] -> Momentum."""
    g = self.graph
    g.store.sync_compute()                       # bf16 mode: compute copy <- fp32 master (one cast launch)
    images, labels = self.iter_train.get_next()
    x, y = self.to_device(images, labels)
    g.begin_step()
    with g.as_default():
      logits_dst = self.helper_dst.calc_logits(None, x) if self.enbl_dst else None
      logits = self.forward_train(x)
      loss, metrics = self.calc_loss(y, logits, self.trainable_vars)
      if self.enbl_dst:
        loss = loss + self.helper_dst.calc_loss(logits, logits_dst)
    self.optimizer.backward(loss)
    lr = self.lrn_rate(self.global_step)
    self.optimizer.weight_decay = g.store.weight_decay
    self.optimizer.compute_gradients()
    self.optimizer.apply_gradients(lr)
    self.global_step += 1
    return lr, loss, metrics

  def train(self):
    """Train a model and periodically produce checkpoint files."""
    self.warm_start(None)
    if FLAGS.enbl_multi_gpu:
      self.bcast_op()
    nb_iters = FLAGS.nb_iters_override or self.nb_iters_train
    time_prev = timer()
    for idx_iter in range(nb_iters):
      lr, loss, metrics = self.train_step()
      if (idx_iter + 1) % FLAGS.summ_step == 0 and self.is_primary_worker('global'):
        torch.cuda.synchronize()
        time_step = timer() - time_prev
        self.__monitor_progress(lr, loss, metrics, idx_iter, time_step)
        time_prev = timer()
      if self.is_primary_worker('global') and (idx_iter + 1) % FLAGS.save_step == 0:
        self.__save_model(is_train=True)
        self.evaluate()
    if self.is_primary_worker('global'):
      self.__save_model(is_train=True)
      self.__restore_model(is_train=False)
      self.__save_model(is_train=False)
      self.evaluate()

  def evaluate(self):
    """Restore a model from the latest checkpoint files and then evaluate it."""
    self.__restore_model(is_train=False)
    return self.run_eval()

  def run_eval(self):
    nb_iters = FLAGS.nb_eval_batches_override or int(np.ceil(float(FLAGS.nb_smpls_eval) / FLAGS.batch_size_eval))
    rslts = []
    self.dump_n_eval(outputs=None, action='init')
    self.iter_eval.reset()
    g = self.graph
    with torch.no_grad():
      for __ in range(nb_iters):
        images, labels = self.iter_eval.get_next()
        x, y = self.to_device(images, labels)
        g.begin_step()
        with g.as_default():
          logits = self.forward_eval(x)
          loss, metrics = self.calc_loss(y, logits, self.trainable_vars)
          if self.enbl_dst:
            loss = loss + self.helper_dst.calc_loss(logits, self.helper_dst.calc_logits(None, x))
        self.dump_n_eval(outputs=logits, action='dump')
        rslts.append([float(loss)] + [float(v) for v in metrics.values()])
        self.eval_op_names = ['loss'] + list(metrics.keys())
    self.dump_n_eval(outputs=None, action='eval')
    means = np.mean(np.array(rslts), axis=0)
    out = {}
    for idx, name in enumerate(self.eval_op_names):
      log.info('%s = %.4e', name, means[idx])
      out[name] = float(means[idx])
    return out

  # ---------------------------------------------------------------------------------------------
  def __save_model(self, is_train):
    if is_train:
      save_path = self.save_vars(FLAGS.save_path, self.global_step)
    else:
      save_path = self.save_vars(FLAGS.save_path_eval)
    log.info('model saved to ' + save_path)

  def __restore_model(self, is_train):
    save_path = checkpoint.latest_checkpoint(os.path.dirname(FLAGS.save_path))
    self.restore_vars(save_path)
    log.info('model restored from ' + save_path)

  def __monitor_progress(self, lr, loss, metrics, idx_iter, time_step):
    speed = FLAGS.batch_size * FLAGS.summ_step / time_step
    if FLAGS.enbl_multi_gpu:
      speed *= mgw.size()
    names = ['lr', 'loss'] + list(metrics.keys())
    vals = [lr, float(loss.detach())] + [float(v) for v in metrics.values()]
    if self.sm_writer is not None:
      self.sm_writer.add_summary(dict(zip(names, vals)), idx_iter)
    log_str = ' | '.join(['%s = %.4e' % (n, v) for n, v in zip(names, vals)])
    log.info('iter #%d: %s | speed = %.2f pics / sec', idx_iter + 1, log_str, speed)
