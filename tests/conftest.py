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


@pytest.fixture(autouse=True)
def _dirty_allocator(request):
  """GPU tests start with NaN patterns in the caching allocator's free blocks, as they would be after a long run:
  a kernel that reads a `torch.empty` buffer it never wrote (or assumes a fresh allocation is zero) then fails in
  every test order, not only when an earlier test happened to leave garbage behind.  PF_TEST_POISON=0 disables."""
  if request.node.get_closest_marker('gpu') is not None and os.environ.get('PF_TEST_POISON', '1') != '0':
    import torch
    if torch.cuda.is_available():
      small = [torch.full((1 << 16,), float('nan'), device='cuda') for _ in range(256)]      # 256 KiB blocks (small pool)
      mid = [torch.full((1 << 20,), float('nan'), device='cuda') for _ in range(64)]         # 4 MiB blocks
      big = [torch.full((1 << 26,), float('nan'), device='cuda') for _ in range(6)]          # 256 MiB blocks
      torch.cuda.synchronize()
      del small, mid, big
  yield


def _tuning_reload():
  """The kernel library reads its PF_* switches once (pf_tuning(), csrc/pf_api.hip): re-read them after an in-process change."""
  mod = sys.modules.get('pocketflow_amd.hip')
  if mod is not None:
    mod.tuning_reload()


@pytest.fixture
def monkeypatch(monkeypatch):
  """pytest's monkeypatch, plus: setenv / delenv of a PF_* switch make the kernel library re-read its switches, and so does the
  teardown (after the environment is restored), so that no test leaves a stale decision behind."""
  class _Patch(object):
    def __getattr__(self, name):
      return getattr(monkeypatch, name)

    def setenv(self, name, value, *a, **kw):
      monkeypatch.setenv(name, value, *a, **kw)
      if name.startswith('PF_'):
        _tuning_reload()

    def delenv(self, name, *a, **kw):
      monkeypatch.delenv(name, *a, **kw)
      if name.startswith('PF_'):
        _tuning_reload()
  yield _Patch()
  monkeypatch.undo()
  _tuning_reload()
