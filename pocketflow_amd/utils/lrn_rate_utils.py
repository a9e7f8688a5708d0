"""Learning-rate schedules (reference utils/lrn_rate_utils.py:23-70).

The reference returns TF tensors evaluated in-graph at `global_step`; here a schedule is a plain
callable `lrn_rate(global_step) -> float` evaluated on the host and passed to the fused optimiser
kernel by value.
"""
from __future__ import annotations

from pocketflow_amd.flags import FLAGS


def piecewise_constant(boundaries, values):
  """tf.train.piecewise_constant: values[0] if x <= b[0]; values[i] if b[i-1] < x <= b[i]; else last."""
  boundaries, values = list(boundaries), list(values)
  if len(values) != len(boundaries) + 1:
    raise ValueError('The length of boundaries should be 1 less than the length of values')

  def lrn_rate(x):
    for b, v in zip(boundaries, values):
      if x <= b:
        return v
    return values[-1]
  return lrn_rate


def setup_lrn_rate_piecewise_constant(global_step, batch_size, idxs_epoch, decay_rates):
  """Piecewise-constant schedule with linear batch-size scaling.  `global_step` is unused (kept for
  signature parity); returns the schedule callable."""
  idxs_epoch = [idx_epoch * FLAGS.nb_epochs_rat for idx_epoch in idxs_epoch]
  lrn_rate_init = FLAGS.lrn_rate_init * batch_size / FLAGS.batch_size_norm
  nb_batches_per_epoch = float(FLAGS.nb_smpls_train) / batch_size
  bnds = [int(nb_batches_per_epoch * idx_epoch) for idx_epoch in idxs_epoch]
  vals = [lrn_rate_init * decay_rate for decay_rate in decay_rates]
  return piecewise_constant(bnds, vals)


def setup_lrn_rate_exponential_decay(global_step, batch_size, epoch_step, decay_rate):
  """Exponential (staircase) decay schedule."""
  epoch_step *= FLAGS.nb_epochs_rat
  lrn_rate_init = FLAGS.lrn_rate_init * batch_size / FLAGS.batch_size_norm
  batch_step = int(FLAGS.nb_smpls_train * epoch_step / batch_size)
  return lambda step: lrn_rate_init * (decay_rate ** (int(step) // batch_step))
