"""Non-uniform quantisation learner (reference learners/nonuniform_quantization/learner.py:34-470).

Same loop as the uniform learner; the weight quantiser is the codebook kernel (pf_seg_nuq_apply) and,
when nuql_opt_mode is 'cluster' or 'both', the codebook gradient kernel (pf_seg_nuq_codebook_grad)
adds dL/dclusters to the flat gradient buffer before the (all-reduce and) Adam launch.  Initialisation
order as in the reference (:120-135): variables -> [warm-start restore] -> cluster_init -> bcast.
"""
from __future__ import annotations

import logging
import os
from timeit import default_timer as timer

import numpy as np
import torch

from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.learners.abstract_learner import AbstractLearner
from pocketflow_amd.learners import teacher_ahead
from pocketflow_amd.learners.distillation_helper import DistillationHelper
from pocketflow_amd.learners.nonuniform_quantization.bit_optimizer import BitOptimizer
from pocketflow_amd.learners.nonuniform_quantization.utils import NonUniformQuantization
from pocketflow_amd.optim import FlatOptimizer
from pocketflow_amd.utils import checkpoint
from pocketflow_amd.utils.lrn_rate_utils import piecewise_constant
from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw

flags.DEFINE_string('nuql_init_style', 'quantile', 'Initialization: quantile(default), uniform]')
flags.DEFINE_string('nuql_opt_mode', 'weights', 'Optimize: weights(default), clusters, both')
flags.DEFINE_integer('nuql_weight_bits', 4, 'Number of bits to use for quantizing weights')
flags.DEFINE_integer('nuql_activation_bits', 32, 'WARNING: Useless for activation quantization in non-uniform mode')
flags.DEFINE_boolean('nuql_use_buckets', False, 'Use bucketing or not')
flags.DEFINE_integer('nuql_bucket_size', 256, 'Number of bucket size')
flags.DEFINE_integer('nuql_quant_epochs', 60, 'Number of finetune steps for quantization')
flags.DEFINE_string('nuql_save_quant_model_path', './nuql_quant_models/model.ckpt', 'dir to save quantization model')
flags.DEFINE_boolean('nuql_quantize_all_layers', False,
                     'True for quantizing all layers and Flase for leaving first and last layers unquantized')
flags.DEFINE_string('nuql_bucket_type', 'split', '[split, channel]')

log = logging.getLogger('pocketflow_amd')


def setup_bnds_decay_rates(model_name, dataset_name):
  """NOTE: The bnd_decay_rates here is mgw_size invariant (reference :52-73); models without a table
   fall back to the cifar_10/resnet row."""
  batch_size = FLAGS.batch_size if not FLAGS.enbl_multi_gpu else FLAGS.batch_size * mgw.size()
  nb_batches_per_epoch = int(FLAGS.nb_smpls_train / batch_size)
  mgw_size = int(mgw.size()) if FLAGS.enbl_multi_gpu else 1
  init_lr = FLAGS.lrn_rate_init * FLAGS.batch_size * mgw_size / FLAGS.batch_size_norm \
      if FLAGS.enbl_multi_gpu else FLAGS.lrn_rate_init
  bnds = [nb_batches_per_epoch * 40, nb_batches_per_epoch * 80]
  decay_rates = [1e-4, 1e-5, 1e-6]
  if dataset_name == 'ilsvrc_12':
    if model_name.startswith('resnet'):
      bnds = [nb_batches_per_epoch * 5, nb_batches_per_epoch * 20]
      decay_rates = [5e-4, 5e-5, 5e-6]
    elif model_name.startswith('mobilenet'):
      bnds = [nb_batches_per_epoch * 5, nb_batches_per_epoch * 30]
      decay_rates = [1e-4, 1e-5, 1e-6]
  finetune_steps = nb_batches_per_epoch * FLAGS.nuql_quant_epochs
  init_lr = init_lr if FLAGS.enbl_warm_start else FLAGS.lrn_rate_init
  return init_lr, bnds, decay_rates, finetune_steps


class NonUniformQuantLearner(AbstractLearner):
  # pylint: disable=too-many-instance-attributes
  """Nonuniform quantization for weights and uniform quantization for activations."""

  def __init__(self, sm_writer, model_helper):
    super(NonUniformQuantLearner, self).__init__(sm_writer, model_helper)
    if FLAGS.enbl_dst:
      self.helper_dst = DistillationHelper(sm_writer, model_helper, self.mpi_comm)
    self.ops = {}
    self.bit_placeholders = {}
    self.statistics = {}
    self.__build_train()
    self.__build_eval()
    if self.is_primary_worker('local'):
      self.download_model()  # pre-trained model is required
    self.auto_barrier()
    self.clusters_initialized = False
    bit_optimizer = BitOptimizer(self.dataset_name, self.weights, self.statistics, self.bit_placeholders, self.ops,
                                 (None, None), self, self, None, None, self.auto_barrier, mpi_comm=self.mpi_comm)
    self.optimal_w_bit_list, self.optimal_a_bit_list = bit_optimizer.run()
    from pocketflow_amd import step_graph
    self.nonuni_quant.on_change = lambda: step_graph.invalidate(self)
    self.nonuni_quant.feed_bits(self.optimal_w_bit_list, self.optimal_a_bit_list)
    self.auto_barrier()

  # ---------------------------------------------------------------------------------------------
  def _train_step_eager(self, optimizer=None):
    """ops['train'] (Adam) or, with `optimizer=self.optimizer_fintune`, ops['rl_fintune'] (SGD) of the reference."""
    optimizer = optimizer or self.optimizer
    g = self.graph
    ahead, x, y, logits_dst = teacher_ahead.next_batch(self)   # batch + teacher logits issued by the previous step on the side stream (PF_TEACHER_AHEAD=0: in line)
    g.begin_step()
    self.nonuni_quant.quantize_weights()
    with g.as_default():
      if FLAGS.enbl_dst and logits_dst is None:
        logits_dst = self.helper_dst.calc_logits(None, x)
      logits = self.forward_train(x)
      model_loss, metrics = self.calc_loss(y, logits, self.trainable_vars)
      loss, dst_loss = model_loss, None
      if FLAGS.enbl_dst:
        dst_loss = self.helper_dst.calc_loss(logits, logits_dst)
        loss = loss + dst_loss
    optimizer.backward(loss)
    if FLAGS.nuql_opt_mode in ('cluster', 'both'):
      self.nonuni_quant.codebook_grads()
    lr = self.lrn_rate(self.ft_step)
    optimizer.weight_decay = g.store.weight_decay
    optimizer.compute_gradients()
    optimizer.apply_gradients(lr)
    self.ft_step += 1
    if ahead is not None:
      ahead.issue()                                 # next batch's teacher forward on the side stream: it runs beside the NEXT step's forward pass
    return {'lr': lr, 'dst_loss': dst_loss, 'model_loss': model_loss, 'loss': loss, 'metrics': metrics}

  # -- callables handed to the bit optimiser (the reference passes TF ops + sessions) --------------------
  def __op_non_cluster_init(self):
    """ops['non_cluster_init']: every variable but the codebooks, optimiser slots, step counter."""
    self.graph.store.initialize(FLAGS.init_seed)
    self.optimizer.reset_slots()
    self.ft_step = 0

  def __op_cluster_init(self, w_bits):
    """ops['cluster_init'] under a bit-width feed: codebooks of 2**bits points from the CURRENT weights."""
    self.nonuni_quant.feed_bits(w_bits, [op.bits for op in self.nonuni_quant.activation_ops])
    self.nonuni_quant.cluster_init()
    self.clusters_initialized = True

  def __log_row(self, r):
    acc_top1, acc_top5 = self.__split_metrics(r['metrics'])
    row = [r['lr']] + ([r['dst_loss']] if FLAGS.enbl_dst else []) + [r['model_loss'], r['loss'], acc_top1, acc_top5]
    return [float(v.detach()) if torch.is_tensor(v) else float(v) for v in row]

  def __op_train(self, w_bits, a_bits, optimizer=None):
    self.nonuni_quant.feed_bits(w_bits, a_bits)        # no-op unless the widths changed (then cluster_init is due)
    return self.__log_row(self.train_step(optimizer))

  def __op_eval(self, w_bits, a_bits):
    self.nonuni_quant.feed_bits(w_bits, a_bits)
    with torch.no_grad():
      self.nonuni_quant.quantize_weights()
      return list(self.__eval_batch())

  def __op_reset_ft_step(self):
    self.ft_step = 0

  def __op_layerwise_tune(self, n, w_bits, a_bits):
    """layerwise_tune_ops[n] + layerwise_diff[n] under a bit-width feed (nuq learner.py:383-385)."""
    from pocketflow_amd.learners.layerwise import LayerwiseTuner, layers_of_vars
    self.nonuni_quant.feed_bits(w_bits, a_bits)
    images = teacher_ahead.next_images(self)             # the batch a previous step prefetched, if any: same data order either way
    if getattr(self, '_layer_tuner', None) is None:
      layers = layers_of_vars(self.graph, self.forward_eval, images, [op.var for op in self.nonuni_quant.matmul_ops])
      self._layer_tuner = LayerwiseTuner(self.graph, self.forward_train, layers)
    return self._layer_tuner.step(n, images, self.nonuni_quant.quantize_weights)

  def init_clusters(self):
    """ops['cluster_init'] (+ bcast) -- after the weights are in place."""
    self.nonuni_quant.cluster_init()
    self.clusters_initialized = True
    if FLAGS.enbl_multi_gpu:
      self.ops['bcast']()

  def train(self):
    total_iters = FLAGS.nb_iters_override or self.finetune_steps
    if FLAGS.enbl_warm_start:
      self.__restore_model(is_train=True)
    # NOTE: initialize the clusters after restore weights
    self.nonuni_quant.feed_bits(self.optimal_w_bit_list, self.optimal_a_bit_list)
    self.init_clusters()
    time_prev = timer()
    for idx_iter in range(total_iters):
      log_rslt = self.train_step()
      if (idx_iter + 1) % FLAGS.summ_step == 0:
        time_prev = self.__monitor_progress(log_rslt, time_prev, idx_iter)
      if (idx_iter + 1) % FLAGS.save_step == 0:
        self.__save_model()
        self.evaluate()
        log.info("Optimal Weight Quantization:{}".format(self.optimal_w_bit_list))
        self.auto_barrier()
    self.__save_model()
    return self.evaluate()

  def evaluate(self):
    if not self.is_primary_worker():
      return None
    self.__restore_model(is_train=False)
    return self.run_eval()

  def run_eval(self):
    losses, acc1, acc5 = [], [], []
    nb_iters = FLAGS.nb_eval_batches_override or int(np.ceil(float(FLAGS.nb_smpls_eval) / FLAGS.batch_size_eval))
    self.iter_eval.reset()
    self.nonuni_quant.feed_bits(self.optimal_w_bit_list, self.optimal_a_bit_list)
    with torch.no_grad():
      self.nonuni_quant.quantize_weights()
      for _ in range(nb_iters):
        loss, a1, a5 = self.__eval_batch()
        losses.append(loss); acc1.append(a1); acc5.append(a5)
    log.info('loss: {}'.format(np.mean(np.array(losses))))
    log.info('accuracy: {}'.format(np.mean(np.array(acc1))))
    return {'loss': float(np.mean(losses)), 'acc_top1': float(np.mean(acc1)), 'acc_top5': float(np.mean(acc5))}

  # ---------------------------------------------------------------------------------------------
  def __eval_batch(self):
    """One run of ops['eval']: quantised forward_eval on the next evaluation batch (weights already quantised)."""
    g = self.graph
    images, labels = self.iter_eval.get_next()
    x, y = self.to_device(images, labels)
    g.begin_step()
    with g.as_default():
      logits = self.forward_eval(x)
      loss, metrics = self.calc_loss(y, logits, self.trainable_vars)
      if FLAGS.enbl_dst:
        loss = loss + self.helper_dst.calc_loss(logits, self.helper_dst.calc_logits(None, x))
    a1, a5 = self.__split_metrics(metrics)
    return float(loss), float(a1), float(a5)

  def __split_metrics(self, metrics):
    if self.dataset_name == 'cifar_10':
      return metrics['accuracy'], 0.0
    elif self.dataset_name == 'ilsvrc_12':
      return metrics['acc_top1'], metrics['acc_top5']
    raise ValueError("Unrecognized dataset name")

  def __declare_quant(self, graph):
    """Scan the freshly declared graph and create the `clusters` variables (before finalize)."""
    nq = NonUniformQuantization(graph, FLAGS.nuql_bucket_size, FLAGS.nuql_use_buckets, FLAGS.nuql_init_style,
                                FLAGS.nuql_bucket_type)
    matmul_ops = nq.search_matmul_op(FLAGS.nuql_quantize_all_layers)
    act_ops = nq.search_activation_op()
    self.statistics['nb_matmuls'] = len(matmul_ops)
    self.statistics['nb_activations'] = len(act_ops)
    self._w_bit_dict = {op.name: FLAGS.nuql_weight_bits for op in matmul_ops}
    self._a_bit_dict = {op.name: FLAGS.nuql_activation_bits for op in act_ops}
    # a bit-width search re-sizes the codebooks every roll-out: allocate them for the widest setting
    nq.declare_clusters(self._w_bit_dict, FLAGS.nuql_w_bit_max if FLAGS.nuql_enbl_rl_agent else 0)
    self.nonuni_quant = nq

  def __build_train(self):
    self.graph = self.build_graph(self.model_scope, separate_compute=True, before_finalize=self.__declare_quant)
    st = self.graph.store
    self.iter_train = self.build_dataset_train().to(self.device)
    self.weights = [v for v in self.trainable_vars if 'kernel' in v.name or 'weight' in v.name]
    if not FLAGS.nuql_quantize_all_layers:
      self.weights = self.weights[1:-1]
    self.statistics['num_weights'] = [v.numel for v in self.weights]
    self.nonuni_quant.insert_quant_op_for_weights(self._w_bit_dict)
    self.nonuni_quant.insert_quant_op_for_activations(self._a_bit_dict)

    self.ft_step = 0
    init_lr, bnds, decay_rates, self.finetune_steps = setup_bnds_decay_rates(self.model_name, self.dataset_name)
    self.lrn_rate = piecewise_constant([i for i in bnds], [init_lr * decay_rate for decay_rate in decay_rates])
    optimizer = FlatOptimizer(st, 'adam')
    # var_list selection (:253-274): masks over the two flat buffers
    clusters = [v for v in self.trainable_vars if 'clusters' in v.name]
    o_mask = torch.ones_like(st.o_master)
    w_mask = None
    if FLAGS.nuql_opt_mode == 'weights':
      for v in clusters:
        o_mask[v.offset:v.offset + v.numel] = 0
    elif FLAGS.nuql_opt_mode == 'cluster':
      o_mask.zero_()
      for v in clusters:
        o_mask[v.offset:v.offset + v.numel] = 1
      w_mask = torch.zeros_like(st.w_master)
    elif FLAGS.nuql_opt_mode != 'both':
      raise ValueError("Unknown optimization mode")
    optimizer.o_mask, optimizer.w_mask = o_mask, w_mask
    # roll-outs of the bit-width search fine-tune with plain SGD when codebooks are optimised (:262-266, 282-285)
    optimizer_fintune = None
    if FLAGS.nuql_opt_mode in ('cluster', 'both') and FLAGS.nuql_enbl_rl_agent:
      optimizer_fintune = FlatOptimizer(st, 'momentum', momentum=0.0)
      optimizer_fintune.o_mask, optimizer_fintune.w_mask = o_mask, w_mask
    if FLAGS.enbl_multi_gpu:
      optimizer = mgw.DistributedOptimizer(optimizer)
      if optimizer_fintune is not None:
        optimizer_fintune = mgw.DistributedOptimizer(optimizer_fintune)
    self.optimizer = optimizer
    self.optimizer_fintune = optimizer_fintune
    self.ops['bcast'] = mgw.broadcast_global_variables(0, [st], [optimizer]) if FLAGS.enbl_multi_gpu else None
    self.ops.update({'non_cluster_init': self.__op_non_cluster_init, 'cluster_init': self.__op_cluster_init,
                     'train': self.__op_train, 'eval': self.__op_eval, 'reset_ft_step': self.__op_reset_ft_step,
                     'layerwise_tune': self.__op_layerwise_tune,
                     'rl_fintune': (lambda w, a: self.__op_train(w, a, self.optimizer_fintune)),
                     'restore': (lambda path: self.restore_vars(path, strict=False)),
                     'save': lambda path: self.save_vars(path)})

  def __build_eval(self):
    self.iter_eval = self.build_dataset_eval().to(self.device)
    self.ops['bucket_storage'] = self.nonuni_quant.bucket_storage

  def __save_model(self):
    if not self.is_primary_worker():
      return
    path = self.save_vars(FLAGS.nuql_save_quant_model_path, self.ft_step)
    log.info('quantized model saved to ' + path)

  def __restore_model(self, is_train):
    if is_train:
      save_path = checkpoint.latest_checkpoint(os.path.dirname(FLAGS.save_path))
      self.restore_vars(save_path, strict=False)     # a full-precision checkpoint has no `clusters`
    else:
      save_path = checkpoint.latest_checkpoint(os.path.dirname(FLAGS.nuql_save_quant_model_path))
      self.restore_vars(save_path)
    log.info('model restored from ' + save_path)

  def __monitor_progress(self, log_rslt, time_prev, idx_iter):
    if not self.is_primary_worker():
      return None
    torch.cuda.synchronize()
    speed = FLAGS.batch_size * FLAGS.summ_step / (timer() - time_prev)
    if FLAGS.enbl_multi_gpu:
      speed *= mgw.size()
    acc_top1, acc_top5 = self.__split_metrics(log_rslt['metrics'])
    dst = ' | dst_loss = %.4f' % float(log_rslt['dst_loss']) if FLAGS.enbl_dst else ''
    log.info('iter #%d: lr = %e%s | model_loss = %.4f | loss = %.4f | acc_top1 = %.4f | acc_top5 = %.4f | '
             'speed = %.2f pics / sec', idx_iter + 1, log_rslt['lr'], dst, float(log_rslt['model_loss']),
             float(log_rslt['loss']), float(acc_top1), float(acc_top5), speed)
    self.last_speed = speed
    return timer()
