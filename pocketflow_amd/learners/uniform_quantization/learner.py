"""Uniform quantisation learner (reference learners/uniform_quantization/learner.py:36-445).

Per step (`sess.run(ops['train'])` in the reference, uq learner.py:133-148):
  data -> teacher forward -> weight fake-quant of all kernels (2 launches) -> student forward with
  fused BN+ReLU+activation-fake-quant -> CE + coupled L2 + distillation (fused kernel) -> backward with
  straight-through estimators -> [RCCL all-reduce of the flat gradient buffers] -> fused Adam.
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
from pocketflow_amd import step_graph
from pocketflow_amd.learners.distillation_helper import DistillationHelper
from pocketflow_amd.learners.uniform_quantization.bit_optimizer import BitOptimizer
from pocketflow_amd.learners.uniform_quantization.utils import UniformQuantization
from pocketflow_amd.optim import FlatOptimizer
from pocketflow_amd.utils import checkpoint
from pocketflow_amd.utils.lrn_rate_utils import piecewise_constant
from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw

flags.DEFINE_integer('uql_weight_bits', 4, 'Number of bits to use for quantizing weights')
flags.DEFINE_integer('uql_activation_bits', 32, 'Number of bits to use for quantizing activations')
flags.DEFINE_boolean('uql_use_buckets', False, 'Use bucketing or not')
flags.DEFINE_integer('uql_bucket_size', 256, 'Number of bucket size')
flags.DEFINE_integer('uql_quant_epochs', 60, 'To be determined by datasets')
flags.DEFINE_string('uql_save_quant_model_path', './uql_quant_models/uql_quant_model.ckpt',
                    'dir to save quantization model')
flags.DEFINE_boolean('uql_quantize_all_layers', False, 'If False, leaving first and last layers unquantized')
flags.DEFINE_string('uql_bucket_type', 'channel', 'Two types for now: [channel, split]')

log = logging.getLogger('pocketflow_amd')


def setup_bnds_decay_rates(model_name, dataset_name):
  """NOTE: The bnd_decay_rates here is mgw_size invariant (reference :50-70).

  Deviation: the reference has no table for models other than resnet*/mobilenet* and raises
  UnboundLocalError for e.g. `lenet`; such models fall back to the cifar_10/resnet row.
  """
  batch_size = FLAGS.batch_size if not FLAGS.enbl_multi_gpu else FLAGS.batch_size * mgw.size()
  nb_batches_per_epoch = int(FLAGS.nb_smpls_train / batch_size)
  mgw_size = int(mgw.size()) if FLAGS.enbl_multi_gpu else 1
  init_lr = FLAGS.lrn_rate_init * FLAGS.batch_size * mgw_size / FLAGS.batch_size_norm \
      if FLAGS.enbl_multi_gpu else FLAGS.lrn_rate_init
  bnds = [nb_batches_per_epoch * 15, nb_batches_per_epoch * 40]
  decay_rates = [1e-3, 1e-4, 1e-5]
  if dataset_name == 'ilsvrc_12':
    if model_name.startswith('resnet'):
      bnds = [nb_batches_per_epoch * 5, nb_batches_per_epoch * 20]
      decay_rates = [1e-4, 1e-5, 1e-6]
    elif model_name.startswith('mobilenet'):
      bnds = [nb_batches_per_epoch * 5, nb_batches_per_epoch * 30]
      decay_rates = [1e-4, 1e-5, 1e-6]
  finetune_steps = nb_batches_per_epoch * FLAGS.uql_quant_epochs
  init_lr = init_lr if FLAGS.enbl_warm_start else FLAGS.lrn_rate_init
  return init_lr, bnds, decay_rates, finetune_steps


class UniformQuantLearner(AbstractLearner):
  # pylint: disable=too-many-instance-attributes
  """Uniform quantization for weights and activations."""

  def __init__(self, sm_writer, model_helper):
    super(UniformQuantLearner, self).__init__(sm_writer, model_helper)

    if FLAGS.enbl_dst:
      self.helper_dst = DistillationHelper(sm_writer, model_helper, self.mpi_comm)

    self.ops = {}
    self.bit_placeholders = {}
    self.statistics = {}

    self.__build_train()  # for train
    self.__build_eval()  # for eval

    if self.is_primary_worker('local'):
      self.download_model()  # pre-trained model is required
    self.auto_barrier()

    # determine the optimal policy (constant bit widths, or the DDPG search over the callables in self.ops)
    bit_optimizer = BitOptimizer(self.dataset_name, self.weights, self.statistics, self.bit_placeholders, self.ops,
                                 self.layerwise_tune_list, self, self, None, None, self.auto_barrier,
                                 mpi_comm=self.mpi_comm)
    self.optimal_w_bit_list, self.optimal_a_bit_list = bit_optimizer.run()
    self.__feed(self.optimal_w_bit_list, self.optimal_a_bit_list)
    self.auto_barrier()

  # ---------------------------------------------------------------------------------------------
  def _train_step_eager(self):
    """ops['train'] of the reference, one iteration."""
    g = self.graph
    ahead, x, y, logits_dst = teacher_ahead.next_batch(self)   # batch + teacher logits issued by the previous step on the side stream (PF_TEACHER_AHEAD=0: in line)
    g.begin_step()
    self.uni_quant.quantize_weights()
    with g.as_default():
      if FLAGS.enbl_dst and logits_dst is None:
        logits_dst = self.helper_dst.calc_logits(None, x)
      logits = self.forward_train(x)
      model_loss, metrics = self.calc_loss(y, logits, self.trainable_vars)
      loss = model_loss
      dst_loss = None
      if FLAGS.enbl_dst:
        dst_loss = self.helper_dst.calc_loss(logits, logits_dst)
        loss = loss + dst_loss
    self.optimizer.backward(loss)
    lr = self.lrn_rate(self.ft_step)
    self.optimizer.weight_decay = g.store.weight_decay
    self.optimizer.compute_gradients()
    self.optimizer.apply_gradients(lr)
    self.ft_step += 1
    if ahead is not None:
      ahead.issue()                                 # next batch's teacher forward on the side stream: it runs beside the NEXT step's forward pass
    return {'lr': lr, 'dst_loss': dst_loss, 'model_loss': model_loss, 'loss': loss, 'metrics': metrics}

  # -- callables handed to the bit optimiser (the reference passes TF ops + sessions) --------------------
  def __feed(self, w_bits, a_bits):
    key = (tuple(int(b) for b in w_bits), tuple(int(b) for b in a_bits))
    if key != getattr(self, '_fed_bits', None):
      self.uni_quant.feed_bits(*key)
      self._fed_bits = key
      step_graph.invalidate(self)                    # a recorded step carries the bit widths by value

  def __op_init(self):
    """ops['init'] = tf.global_variables_initializer(): fresh variables, empty Adam slots, step 0."""
    self.graph.store.initialize(FLAGS.init_seed)
    self.optimizer.reset_slots()
    self.ft_step = 0

  def __op_train(self, w_bits, a_bits):
    """[ops['train'], ops['log']] under a bit-width feed: one quantisation-aware step, returns the log row."""
    self.__feed(w_bits, a_bits)
    r = self.train_step()
    acc_top1, acc_top5 = self.__split_metrics(r['metrics'])
    row = [r['lr']] + ([r['dst_loss']] if FLAGS.enbl_dst else []) + [r['model_loss'], r['loss'], acc_top1, acc_top5]
    return [float(v.detach()) if torch.is_tensor(v) else float(v) for v in row]

  def __op_eval(self, w_bits, a_bits):
    """ops['eval'] under a bit-width feed: [loss, acc_top1, acc_top5] of the next evaluation batch."""
    self.__feed(w_bits, a_bits)
    with torch.no_grad():
      self.uni_quant.quantize_weights()
      return list(self.__eval_batch())

  def __op_reset_ft_step(self):
    self.ft_step = 0

  def __op_layerwise_tune(self, n, w_bits, a_bits):
    """layerwise_tune_ops[n] + layerwise_diff[n] under a bit-width feed (uq utils.py:136-161)."""
    from pocketflow_amd.learners.layerwise import LayerwiseTuner, layers_of_vars
    self.__feed(w_bits, a_bits)
    images = teacher_ahead.next_images(self)             # the batch a previous step prefetched, if any: same data order either way
    if getattr(self, '_layer_tuner', None) is None:
      layers = layers_of_vars(self.graph, self.forward_eval, images, [op.var for op in self.uni_quant.matmul_ops])
      self._layer_tuner = LayerwiseTuner(self.graph, self.forward_train, layers)
    return self._layer_tuner.step(n, images, self.uni_quant.quantize_weights)

  def train(self):
    total_iters = FLAGS.nb_iters_override or self.finetune_steps
    if FLAGS.enbl_warm_start:
      self.__restore_model(is_train=True)  # use the latest model for warm start
    self.auto_barrier()
    if FLAGS.enbl_multi_gpu:
      self.ops['bcast']()
    time_prev = timer()
    self.__feed(self.optimal_w_bit_list, self.optimal_a_bit_list)
    for idx_iter in range(total_iters):
      log_rslt = self.train_step()
      if (idx_iter + 1) % FLAGS.summ_step == 0:
        time_prev = self.__monitor_progress(log_rslt, time_prev, idx_iter)
      if (idx_iter + 1) % FLAGS.save_step == 0:
        self.__save_model()
        self.evaluate()
        self.auto_barrier()
    self.__save_model()
    return self.evaluate()

  def evaluate(self):
    if not self.is_primary_worker():
      return None
    self.__restore_model(is_train=False)
    return self.run_eval()

  def run_eval(self):
    """sess_eval.run(ops['eval']) over the evaluation subset: mean of per-batch [loss, top1, top5]."""
    losses, acc1, acc5 = [], [], []
    nb_iters = FLAGS.nb_eval_batches_override or int(np.ceil(float(FLAGS.nb_smpls_eval) / FLAGS.batch_size_eval))
    self.iter_eval.reset()
    self.__feed(self.optimal_w_bit_list, self.optimal_a_bit_list)
    with torch.no_grad():
      self.uni_quant.quantize_weights()          # re-quantise the restored fp32 shadows (App. A.8)
      for _ in range(nb_iters):
        loss, a1, a5 = self.__eval_batch()
        losses.append(loss); acc1.append(a1); acc5.append(a5)
    log.info('loss: {}'.format(np.mean(np.array(losses))))
    log.info('accuracy: {}'.format(np.mean(np.array(acc1))))
    log.info("Optimal Weight Quantization:{}".format(self.optimal_w_bit_list))
    if FLAGS.uql_use_buckets:
      self.__show_bucket_storage(self.uni_quant.bucket_storage)
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

  def __build_train(self):
    # model definition (variables + ops), data pipeline
    self.graph = self.build_graph(self.model_scope, separate_compute=True)
    self.iter_train = self.build_dataset_train().to(self.device)
    self.weights = [v for v in self.trainable_vars if 'kernel' in v.name or 'weight' in v.name]
    if not FLAGS.uql_quantize_all_layers:
      self.weights = self.weights[1:-1]
    self.statistics['num_weights'] = [v.numel for v in self.weights]

    self.__quantize_train_graph()

    # optimizer & gradients
    self.ft_step = 0
    init_lr, bnds, decay_rates, self.finetune_steps = setup_bnds_decay_rates(self.model_name, self.dataset_name)
    self.lrn_rate = piecewise_constant([i for i in bnds], [init_lr * decay_rate for decay_rate in decay_rates])
    optimizer = FlatOptimizer(self.graph.store, 'adam')
    if FLAGS.enbl_multi_gpu:
      optimizer = mgw.DistributedOptimizer(optimizer)
    self.optimizer = optimizer
    self.ops['bcast'] = mgw.broadcast_global_variables(0, [self.graph.store], [optimizer]) \
        if FLAGS.enbl_multi_gpu else None
    self.ops.update({'init': self.__op_init, 'train': self.__op_train, 'eval': self.__op_eval,
                     'reset_ft_step': self.__op_reset_ft_step, 'restore': self.restore_vars,
                     'layerwise_tune': self.__op_layerwise_tune,
                     'save': lambda path: self.save_vars(path)})

  def __build_eval(self):
    self.iter_eval = self.build_dataset_eval().to(self.device)
    self.__quantize_eval_graph()

  def __quantize_train_graph(self):
    """Insert quantization nodes to the training graph."""
    uni_quant = UniformQuantization(self.graph, FLAGS.uql_bucket_size, FLAGS.uql_use_buckets,
                                    FLAGS.uql_bucket_type)
    matmul_ops = uni_quant.search_matmul_op(FLAGS.uql_quantize_all_layers)
    act_ops = uni_quant.search_activation_op()
    self.statistics['nb_matmuls'] = len(matmul_ops)
    self.statistics['nb_activations'] = len(act_ops)
    matmul_op_names = [op.name for op in matmul_ops]
    act_op_names = [op.name for op in act_ops]
    self.bit_placeholders['w_train'] = [FLAGS.uql_weight_bits] * len(matmul_ops)
    self.bit_placeholders['a_train'] = [FLAGS.uql_activation_bits] * len(act_ops)
    w_bit_dict_train = self.__build_quant_dict(matmul_op_names, self.bit_placeholders['w_train'])
    a_bit_dict_train = self.__build_quant_dict(act_op_names, self.bit_placeholders['a_train'])
    uni_quant.insert_quant_op_for_weights(w_bit_dict_train)
    uni_quant.insert_quant_op_for_activations(a_bit_dict_train)
    self.uni_quant = uni_quant
    self.layerwise_tune_list = (None, None)

  def __quantize_eval_graph(self):
    """The eval graph shares ops and variables with the train graph; only the counts are checked."""
    assert self.statistics['nb_matmuls'] == len(self.uni_quant.matmul_ops), \
        'the length of matmul_ops on train and eval graphs does not match'
    assert self.statistics['nb_activations'] == len(self.uni_quant.activation_ops), \
        'the length of act_ops on train and eval graphs does not match'
    self.ops['bucket_storage'] = self.uni_quant.bucket_storage

  def __save_model(self):
    if not self.is_primary_worker():
      return
    path = self.save_vars(FLAGS.uql_save_quant_model_path, self.ft_step)
    log.info('quantized model saved to ' + path)

  def __restore_model(self, is_train):
    if is_train:
      save_path = checkpoint.latest_checkpoint(os.path.dirname(FLAGS.save_path))
    else:
      save_path = checkpoint.latest_checkpoint(os.path.dirname(FLAGS.uql_save_quant_model_path))
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
    if self.sm_writer is not None:
      self.sm_writer.add_summary({'loss': float(log_rslt['loss'])}, idx_iter)
    if FLAGS.enbl_dst:
      log.info('iter #%d: lr = %e | dst_loss = %.4f | model_loss = %.4f | loss = %.4f | acc_top1 = %.4f | '
               'acc_top5 = %.4f | speed = %.2f pics / sec', idx_iter + 1, log_rslt['lr'],
               float(log_rslt['dst_loss']), float(log_rslt['model_loss']), float(log_rslt['loss']),
               float(acc_top1), float(acc_top5), speed)
    else:
      log.info('iter #%d: lr = %e | model_loss = %.4f | loss = %.4f | acc_top1 = %.4f | acc_top5 = %.4f | '
               'speed = %.2f pics / sec', idx_iter + 1, log_rslt['lr'], float(log_rslt['model_loss']),
               float(log_rslt['loss']), float(acc_top1), float(acc_top5), speed)
    self.last_speed = speed
    return timer()

  def __show_bucket_storage(self, bucket_storage):
    weight_storage = sum(self.statistics['num_weights']) * FLAGS.uql_weight_bits \
        if not FLAGS.uql_enbl_rl_agent else sum(self.statistics['num_weights']) * FLAGS.uql_equivalent_bits
    log.info('bucket storage: %d bit / %.3f kb | weight storage: %d bit / %.3f kb | ratio: %.3f',
             bucket_storage, bucket_storage / (8. * 1024.), weight_storage, weight_storage / (8. * 1024.),
             bucket_storage * 1. / weight_storage)

  @staticmethod
  def __build_quant_dict(keys, values):
    """Bind op names and bit widths to a dictionary."""
    return {v: values[idx] for idx, v in enumerate(keys)}
