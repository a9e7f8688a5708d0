"""Model helper for ResNet-v2 on CIFAR-10 (reference nets/resnet_at_cifar10.py:29-136):
resnet_size 20 -> 3 stages x (size-2)//6 building blocks, 16 filters, 3x3 stem, no first pool;
loss = CE + loss_w_dcy * L2 over non-BN trainables; 250 epochs, LR x0.1 at 100/150/200."""
import torch

from pocketflow_amd import losses
from pocketflow_amd.datasets.cifar10_dataset import Cifar10Dataset
from pocketflow_amd.flags import FLAGS, flags
from pocketflow_amd.graph import get_default_graph
from pocketflow_amd.nets.abstract_model_helper import AbstractModelHelper
from pocketflow_amd.utils.external import resnet_model as ResNet
from pocketflow_amd.utils.lrn_rate_utils import setup_lrn_rate_piecewise_constant
from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw

flags.DEFINE_integer('resnet_size', 20, '# of layers in the ResNet model')
flags.DEFINE_float('nb_epochs_rat', 1.0, '# of training epochs\'s ratio')
flags.DEFINE_float('lrn_rate_init', 1e-1, 'initial learning rate')
flags.DEFINE_float('batch_size_norm', 128, 'normalization factor of batch size')
flags.DEFINE_float('momentum', 0.9, 'momentum coefficient')
flags.DEFINE_float('loss_w_dcy', 2e-4, 'weight decaying loss\'s coefficient')


def forward_fn(inputs, is_train, data_format):
  graph = get_default_graph()
  net = graph.nets.get('resnet')
  if net is None:
    nb_blocks = (FLAGS.resnet_size - 2) // 6
    net = graph.nets['resnet'] = ResNet.Model(
        FLAGS.resnet_size, False, FLAGS.nb_classes, 16, 3, 1, None, None, [nb_blocks] * 3, [1, 2, 2],
        data_format=data_format, graph=graph)
  if inputs.device.type == 'meta':
    return torch.empty((inputs.shape[0], FLAGS.nb_classes), device='meta')
  return net(inputs, is_train)


class ModelHelper(AbstractModelHelper):
  """Model helper for creating a ResNet model for the CIFAR-10 dataset."""

  def __init__(self, data_format='channels_last'):
    super(ModelHelper, self).__init__(data_format)
    self.dataset_train = Cifar10Dataset(is_train=True)
    self.dataset_eval = Cifar10Dataset(is_train=False)

  def build_dataset_train(self, enbl_trn_val_split=False):
    return self.dataset_train.build(enbl_trn_val_split)

  def build_dataset_eval(self):
    return self.dataset_eval.build()

  def forward_train(self, inputs):
    return forward_fn(inputs, is_train=True, data_format=self.data_format)

  def forward_eval(self, inputs):
    return forward_fn(inputs, is_train=False, data_format=self.data_format)

  def calc_loss(self, labels, outputs, trainable_vars):
    loss = losses.softmax_cross_entropy(labels, outputs)
    loss_filter = lambda var: 'batch_normalization' not in var.name
    loss = loss + losses.l2_regularization(trainable_vars, loss_filter, FLAGS.loss_w_dcy)
    accuracy = (labels.argmax(dim=1) == outputs.argmax(dim=1)).float().mean()
    metrics = {'accuracy': accuracy}
    return loss, metrics

  def setup_lrn_rate(self, global_step):
    nb_epochs = 250
    idxs_epoch = [100, 150, 200]
    decay_rates = [1.0, 0.1, 0.01, 0.001]
    batch_size = FLAGS.batch_size * (1 if not FLAGS.enbl_multi_gpu else mgw.size())
    lrn_rate = setup_lrn_rate_piecewise_constant(global_step, batch_size, idxs_epoch, decay_rates)
    nb_iters = int(FLAGS.nb_smpls_train * nb_epochs * FLAGS.nb_epochs_rat / batch_size)
    return lrn_rate, nb_iters

  @property
  def model_name(self):
    return 'resnet_%d' % FLAGS.resnet_size

  @property
  def dataset_name(self):
    return 'cifar_10'
