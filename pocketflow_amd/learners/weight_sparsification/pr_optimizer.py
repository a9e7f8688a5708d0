"""Pruning-ratio optimizer of the weight-sparsification learner (reference pr_optimizer.py:96-144,
385-409).  'uniform' and 'heurist' are on the hot path; 'optimal' (DDPG roll-outs, :411-611) is a
SURVEY 8f "next" row."""
from __future__ import annotations

import logging

import numpy as np
import torch

from pocketflow_amd.flags import FLAGS
from pocketflow_amd.learners.weight_sparsification.utils import get_maskable_vars
from pocketflow_amd.utils.misc_utils import is_primary_worker

log = logging.getLogger('pocketflow_amd')


class PROptimizer(object):  # pylint: disable=too-many-instance-attributes
  """Pruning ratio optimizer for the weight sparsification learner."""

  def __init__(self, model_helper, mpi_comm):
    self.model_name = model_helper.model_name
    self.dataset_name = model_helper.dataset_name
    self.mpi_comm = mpi_comm
    self.data_scope = 'data'
    self.model_scope_full = 'model'
    self.model_scope_prnd = 'pruned_model'
    if FLAGS.ws_prune_ratio_prtl in ['uniform', 'heurist']:
      self.__build_minimal(model_helper)
    elif FLAGS.ws_prune_ratio_prtl == 'optimal':
      raise NotImplementedError('ws_prune_ratio_prtl=optimal (DDPG) is outside the MI355X hot path; '
                                'use uniform or heurist (SURVEY 8f row 2)')
    else:
      raise ValueError('unrecognzed WS pruning ratio protocol: ' + FLAGS.ws_prune_ratio_prtl)

  def run(self):
    """Return the list of (variable name, pruning ratio) pairs of all maskable variables."""
    if FLAGS.ws_prune_ratio_prtl == 'uniform':
      var_names_n_prune_ratios = self.__calc_uniform_prune_ratios()
    else:
      var_names_n_prune_ratios = self.__calc_heurist_prune_ratios()
    if is_primary_worker('global'):
      for var_name, prune_ratio in var_names_n_prune_ratios:
        log.info('%s: %f' % (var_name, prune_ratio))
    return var_names_n_prune_ratios

  def __build_minimal(self, model_helper):
    """Declare the model on a CPU-side graph only to enumerate its variables (no device work)."""
    from pocketflow_amd.graph import Graph
    from pocketflow_amd.learners.abstract_learner import input_spec
    graph = Graph(self.model_scope_full, 'cpu', torch.float32)
    with graph.as_default():
      model_helper.forward_train(input_spec(model_helper))
    self.vars_full = {'maskable': get_maskable_vars(graph.store.trainable_vars)}

  def __calc_uniform_prune_ratios(self):
    return [(var.name, FLAGS.ws_prune_ratio) for var in self.vars_full['maskable']]

  def __calc_heurist_prune_ratios(self):
    nb_params = np.array([var.numel for var in self.vars_full['maskable']])
    alpha = FLAGS.ws_prune_ratio * np.sum(nb_params) / np.sum(nb_params * np.log(nb_params))
    return [(var.name, alpha * np.log(nb_params[idx])) for idx, var in enumerate(self.vars_full['maskable'])]
