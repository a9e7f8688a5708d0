import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a ROCm GPU (MI355X); run with `-m gpu`')


@pytest.fixture(autouse=True)
def _reset_flags():
  """FLAGS is process-global (like tf.app.flags): restore defaults after every test."""
  yield
  try:
    from pocketflow_amd.flags import FLAGS
    FLAGS.reset()
  except Exception:
    pass
