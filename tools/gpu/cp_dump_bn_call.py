"""One BN backward call of the masked MobileNet fine-tune (Momentum, float32), one channel: inputs and outputs saved.
usage: [PF_HIP_LIB=...] python tools/gpu/cp_dump_bn_call.py <out.npz> <call> <channel>"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import parity_common as PC
import pocketflow_amd.graph as G
from pocketflow_amd.flags import FLAGS
import pocketflow_amd.learners.learner_utils  # noqa
import pocketflow_amd.learners.abstract_learner  # noqa
import pocketflow_amd.nets.mobilenet_at_ilsvrc12  # noqa
import pocketflow_amd.learners.channel_pruning.learner  # noqa
import pocketflow_amd.datasets.abstract_dataset  # noqa
want, ch = int(sys.argv[2]), int(sys.argv[3])
out, n = {}, [0]
orig = G._bn_backward


def rows_of(t, rows, C):
  return (t.permute(0, 2, 3, 1) if t.dim() == 4 else t).reshape(rows, C)


def bwd(dq, x, scale_shift, mean_invstd, act, graph, rows, C, addend=None, params=None, pre=None, **k):
  r = orig(dq, x, scale_shift, mean_invstd, act, graph, rows, C, addend=addend, params=params, pre=pre, **k)
  if n[0] == want:
    torch.cuda.synchronize()
    out['dq'] = rows_of(dq, rows, C)[:, ch].float().cpu().numpy()
    out['x'] = rows_of(x, rows, C)[:, ch].float().cpu().numpy()
    out['ss'] = scale_shift[:, ch].cpu().numpy(); out['mi'] = mean_invstd[:, ch].cpu().numpy()
    out['dx'] = rows_of(r[0], rows, C)[:, ch].float().cpu().numpy()
    out['act'] = np.array([str(act)]); out['rows_C'] = np.array([rows, C])
    out['has_pre'] = np.array([pre is not None]); out['has_addend'] = np.array([addend is not None])
    if pre is not None:
      out['pre'] = pre[0].reshape(-1)[: pre[1] * 2 * C].reshape(pre[1], 2, C)[:, :, ch].cpu().numpy()
    if params is not None:
      gv, bv = G._grad_view(params[0], C), G._grad_view(params[1], C)
      if gv is not None:
        out['dgamma'] = gv[ch:ch + 1].cpu().numpy(); out['dbeta'] = bv[ch:ch + 1].cpu().numpy()
  n[0] += 1
  return r


G._bn_backward = bwd
FLAGS.reset()
with tempfile.TemporaryDirectory() as d:
  tmp_path = pathlib.Path(d)
  FLAGS.save_path = str(tmp_path / 'models' / 'model.ckpt')
  FLAGS.save_path_eval = str(tmp_path / 'models_eval' / 'model.ckpt')
  FLAGS.synthetic_pool = 2
  FLAGS.compute_dtype = 'float32'
  PC.run_cp_masked_finetune(FLAGS, tmp_path, 'momentum', steps=3, report=[])
np.savez_compressed(sys.argv[1], **out)
print('saved', sorted(out))
