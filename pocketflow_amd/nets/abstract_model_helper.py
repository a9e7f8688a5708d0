"""The plugin interface every model implements -- same methods, arity and meaning as the
reference's AbstractModelHelper (nets/abstract_model_helper.py:22-149).

Differences forced by the runtime: tensors are torch tensors (images arrive as float32 NHWC and
are viewed as channels_last), "building TF ops" becomes declaring layers on the default
`pocketflow_amd.graph.Graph` (forward functions are called once in build mode with a `meta` tensor,
then every step for real), `sess` arguments are accepted and ignored, and `setup_lrn_rate` returns a
host-side schedule callable `lrn_rate(global_step) -> float` instead of a TF tensor.
"""
from abc import ABC, abstractmethod


class AbstractModelHelper(ABC):
  """Abstract class for model helpers: data input pipeline & network's forward pass definition."""

  def __init__(self, data_format, forward_w_labels=False):
    """Note: DO NOT declare any variables / layers here (they belong to a Graph)."""
    self.data_format = data_format
    self.forward_w_labels = forward_w_labels

  @abstractmethod
  def build_dataset_train(self, enbl_trn_val_split):
    """Build the data subset for training -> iterator (or (iterator_trn, iterator_val))."""

  @abstractmethod
  def build_dataset_eval(self):
    """Build the data subset for evaluation -> iterator."""

  @abstractmethod
  def forward_train(self, inputs, labels=None):
    """Forward computation at training (inputs -> outputs) on the default Graph."""

  @abstractmethod
  def forward_eval(self, inputs):
    """Forward computation at evaluation."""

  @abstractmethod
  def calc_loss(self, labels, outputs, trainable_vars):
    """Return (loss, metrics: dict); `loss` must be a scalar, differentiable w.r.t. `outputs`."""

  @abstractmethod
  def setup_lrn_rate(self, global_step):
    """Return (lrn_rate schedule, nb_iters)."""

  def warm_start(self, sess):
    """Initialize the model for warm-start (optional)."""

  def dump_n_eval(self, outputs, action):
    """Dump the model's outputs to files and evaluate (optional)."""

  @property
  @abstractmethod
  def model_name(self):
    """Model's name."""

  @property
  @abstractmethod
  def dataset_name(self):
    """Dataset's name."""
