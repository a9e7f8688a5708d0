"""Utility functions of the weight-sparsification learner (reference weight_sparsification/utils.py:19-39)."""


def get_maskable_vars(trainable_vars):
  """Variables that may be masked: convolution / dense kernels (`kernel`), MobileNet's point-wise
  convolutions (`pointwise/weights`) and its final 1x1 logits convolution (`Conv2d_1c_1x1/weights`)."""
  vars_kernel = [var for var in trainable_vars if 'kernel' in var.name]
  vars_ptconv = [var for var in trainable_vars if 'pointwise/weights' in var.name]
  vars_fnconv = [var for var in trainable_vars if 'Conv2d_1c_1x1/weights' in var.name]
  return vars_kernel + vars_ptconv + vars_fnconv
