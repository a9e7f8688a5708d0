"""Bit-allocation search helper of the non-uniform learner (reference learners/nonuniform_quantization/rl_helper.py:25-118):
the uniform learner's helper with the `nuql_` flags."""
from pocketflow_amd.learners.uniform_quantization.rl_helper import RLHelper as _UqRLHelper


class RLHelper(_UqRLHelper):
  FLAG_PREFIX = 'nuql'
