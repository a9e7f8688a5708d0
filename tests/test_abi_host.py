"""CPU-side checks: the C-ABI library loads and exports exactly what include/pocketflow_hip.h declares
(no compute calls -- there is no GPU here), the header is plain C, the host-side launch plans and the
variable store are consistent, and the host logic agrees with fixtures produced by the reference's own
functions (tests/golden/reference_host.json)."""
import ctypes
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'pocketflow_hip.h')


def _declared_symbols():
  src = open(HEADER).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(pf_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
  from pocketflow_amd import hip
  declared = _declared_symbols()
  assert len(declared) >= 30
  lib = ctypes.CDLL(hip.lib_path())
  missing = [s for s in declared if not hasattr(lib, s)]
  assert not missing, 'declared in the header but not exported: %s' % missing
  assert sorted(hip.SYMBOLS) == declared, 'ctypes binding and header disagree: %s' % (
      sorted(set(hip.SYMBOLS) ^ set(declared)))
  assert hip.version() >= 1
  assert b'invalid' in hip._lib.pf_error_string(1).lower() or len(hip._lib.pf_error_string(1)) > 0


def test_host_side_shape_rules_of_the_convolution_entry_points():
  """The pure host functions of the C ABI that tell a caller whether / how an entry point takes a shape (no device work):
  what the layer executor relies on when it chooses between our kernels and the library fallback."""
  from pocketflow_amd import hip
  # the stem kernel is exactly the 7x7 / stride 2 / pad 3, 3 -> 64 convolution with an even height and Wd % 32 == 0
  assert hip.conv_stem_supported(224, 224, 3, 64, 7, 2, 3) and hip.conv_stem_supported(38, 96, 3, 64, 7, 2, 3)
  for bad in ((224, 224, 4, 64, 7, 2, 3), (224, 224, 3, 32, 7, 2, 3), (224, 224, 3, 64, 3, 2, 3), (224, 224, 3, 64, 7, 1, 3),
              (224, 224, 3, 64, 7, 2, 2), (225, 224, 3, 64, 7, 2, 3), (224, 200, 3, 64, 7, 2, 3), (224, 2048, 3, 64, 7, 2, 3)):
    assert not hip.conv_stem_supported(*bad), bad
  # its backward-filter: one slab per persistent workgroup (<= 768), none for rows wider than 256 pixels
  assert hip.conv_stem_wrw_slabs(256, 224, 224) == 768 and hip.conv_stem_wrw_slabs(1, 32, 32) == 8
  assert hip.conv_stem_wrw_slabs(4, 224, 512) == 0 and hip.conv_stem_wrw_slabs(0, 224, 224) == 0
  # RxS backward-filter: C, N multiples of 64 and at least 2048 output pixels; 0 = "use the fallback"
  assert hip.conv2d_wrw_splits(256 * 56 * 56, 64, 64, 9) > 0 and hip.conv2d_wrw_splits(256 * 14 * 14, 256, 256, 9) > 0
  assert hip.conv2d_wrw_splits(1024, 64, 64, 9) == 0 and hip.conv2d_wrw_splits(256 * 56 * 56, 64, 48, 9) == 0
  # statistics groups of the fused forward kernels: at least one, bounded (the consumer BN reduces [G][4][N])
  for M, N, K in ((256 * 56 * 56, 256, 64), (256 * 7 * 7, 2048, 512), (4096, 64, 64)):
    for pro in (False, True):
      assert 1 <= hip.conv1x1_stats_groups(M, N, K, prologue=pro) <= 1024
  import pytest as _pt
  with _pt.raises(ValueError):
    hip.conv2d_stats_groups(256 * 56 * 56, 64)                          # the geometry is mandatory: the kernel depends on it
  for geom in ((256, 56, 56, 64, 64, 3, 3, 1, 1, 1, 56, 56), (256, 14, 14, 256, 256, 3, 3, 1, 1, 1, 14, 14), (2, 56, 56, 64, 64, 3, 3, 1, 1, 1, 56, 56)):
    assert 1 <= hip.conv2d_stats_groups(geom[0] * geom[10] * geom[11], geom[4], geom=geom) <= 1024
  assert hip.conv2d_stats_groups(2 * 56 * 56, 64, geom=(2, 56, 56, 64, 64, 3, 3, 1, 1, 1, 56, 56)) == 56      # window kernel: one row per workgroup = tile


def test_header_is_plain_c_and_struct_sizes_match(tmp_path):
  from pocketflow_amd import hip
  src = tmp_path / 'sz.c'
  src.write_text('#include <stdio.h>\n#include "pocketflow_hip.h"\n'
                 'int main(void){printf("%zu %zu %d\\n", sizeof(PfSeg), sizeof(PfBlock), PF_CHUNK);return 0;}\n')
  exe = tmp_path / 'sz'
  subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
  seg, blk, chunk = map(int, subprocess.check_output([str(exe)]).split())
  assert seg == hip.SEG_DTYPE.itemsize == 64 and blk == hip.BLOCK_DTYPE.itemsize == 16 and chunk == hip.PF_CHUNK


def test_product_never_imports_the_oracle():
  bad = []
  for d, _, files in os.walk(os.path.join(ROOT, 'pocketflow_amd')):
    for f in files:
      if f.endswith('.py'):
        txt = open(os.path.join(d, f)).read()
        if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M):
          bad.append(os.path.join(d, f))
  assert not bad, bad


def test_learners_fail_loudly_without_gpu(tmp_path):
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd.nets.lenet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.full_precision.learner import FullPrecLearner
  FLAGS.save_path = str(tmp_path / 'm' / 'model.ckpt')
  with pytest.raises(RuntimeError, match='need a ROCm GPU'):
    FullPrecLearner(None, ModelHelper())
  from pocketflow_amd import hip
  with pytest.raises(RuntimeError, match='GPU only'):
    hip.minmax_tensor(torch.zeros(8), torch.zeros(2, dtype=torch.int32))


# -- variable store / launch plans ---------------------------------------------------------------------

def _resnet20_graph():
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd.graph import Graph
  from pocketflow_amd.nets.resnet_at_cifar10 import ModelHelper
  from pocketflow_amd.learners.abstract_learner import input_spec
  FLAGS.resnet_size, FLAGS.nb_classes, FLAGS.batch_size = 20, 10, 4
  mh = ModelHelper()
  g = Graph('model', 'cpu', torch.float32)
  with g.as_default():
    mh.forward_train(input_spec(mh))
  g.finalize(seed=1, requires_grad=False)
  return g


def test_varstore_layout_and_reference_layout_roundtrip():
  g = _resnet20_graph()
  st = g.store
  assert len(g.matmul_ops) == 23 and len(g.activation_ops) == 19        # SURVEY 8a row a1 (before [1:-1])
  assert sum(v.numel for v in st.trainable_vars) == 272538     # ResNet-20 trainables (kernels 271152 + BN + dense bias)
  spans = sorted((v.offset, v.offset + v.numel) for v in st.vars if v.group == 'W')
  for (a0, a1), (b0, _) in zip(spans, spans[1:]):
    assert a1 <= b0 and b0 % 64 == 0                                     # disjoint, 256-byte aligned
  vals = st.export_numpy()
  k = vals['model/resnet_model/conv2d_1/kernel']
  assert k.shape[:2] in ((3, 3), (1, 1)) and k.ndim == 4                  # HWIO in the reference layout
  v = st.by_name['model/resnet_model/conv2d_1/kernel']
  assert np.array_equal(v.master.numpy(), np.transpose(k, (3, 0, 1, 2)))  # KRSC storage
  vals2 = {n: a + 1 for n, a in vals.items()}
  st.load_numpy(vals2)
  for n, a in st.export_numpy().items():
    assert np.array_equal(a, vals2[n]), n
  # L2-regularised variables come first in each flat buffer (one n_decay splits them)
  for v in st.vars:
    if v.group == 'O':
      assert (v.offset < st.o_decay) == bool(v.l2), v.name


@pytest.mark.parametrize('use_buckets,bucket_type,bucket_size', [(False, 'channel', 0), (True, 'channel', 0), (True, 'split', 256)])
def test_quant_plan_blocks_cover_every_element_once(use_buckets, bucket_type, bucket_size):
  from pocketflow_amd import hip
  from pocketflow_amd.plan import QuantPlan
  g = _resnet20_graph()
  vs = g.store.matmul_vars
  plan = QuantPlan(g.store.weight_descs(vs), [8] * len(vs), use_buckets, bucket_type, bucket_size, 'cpu')
  segs = plan.segs_host
  blocks = np.frombuffer(plan.ap_blocks.numpy().tobytes(), dtype=hip.BLOCK_DTYPE)
  cover = {s: np.zeros(int(segs[s]['len']), np.int32) for s in range(len(vs))}
  for b in blocks:
    s, c = int(b['seg']), int(b['chunk'])
    e0 = c * hip.PF_CHUNK
    e1 = min(e0 + hip.PF_CHUNK, int(segs[s]['len']))
    cover[s][e0:e1] += 1
    if segs[s]['mode'] == hip.PF_BUCKET_CHANNEL:
      L = int(segs[s]['RS']) * int(segs[s]['I'])
      assert b['row0'] == e0 // L and b['row0'] + b['nrows'] - 1 == (e1 - 1) // L
  assert all(np.all(c == 1) for c in cover.values())
  nb = sum(int(x) for x in segs['n_bucket'])
  assert plan.n_slots == nb
  if use_buckets:
    assert plan.bucket_storage_bits == nb * 64                         # uq utils.py:299-306
  offs = segs['slot_offset']
  assert np.array_equal(offs, np.concatenate([[0], np.cumsum(segs['n_bucket'])[:-1]]))


# -- host logic against the reference's own functions ------------------------------------------------------

@pytest.fixture(scope='module')
def host():
  with open(os.path.join(ROOT, 'tests', 'golden', 'reference_host.json')) as f:
    return json.load(f)


def test_get_path_args_matches_reference_script(host):
  from pocketflow_amd.utils.get_path_args import get_path_args
  conf = os.path.join(ROOT, 'tests', 'golden', 'path.conf.sample')
  for r in host['get_path_args']:
    assert get_path_args(r['mode'], r['run'], conf) == r['stdout']


def test_lrn_rate_utils_match_reference(host):
  from pocketflow_amd.flags import FLAGS
  import pocketflow_amd.nets.resnet_at_cifar10  # noqa: F401  (defines the flags)
  from pocketflow_amd.utils.lrn_rate_utils import setup_lrn_rate_piecewise_constant
  for r in host['lrn_rate_piecewise']:
    FLAGS.nb_smpls_train, FLAGS.lrn_rate_init = r['nb_smpls_train'], r['lrn_rate_init']
    FLAGS.batch_size_norm, FLAGS.nb_epochs_rat = r['batch_size_norm'], r['nb_epochs_rat']
    fn = setup_lrn_rate_piecewise_constant(0, r['batch_size'], r['idxs_epoch'], r['decay_rates'])
    assert np.float32(fn(r['step'])) == np.float32(r['lrn_rate']), r


def test_setup_bnds_decay_rates_match_reference(host, monkeypatch):
  from pocketflow_amd.flags import FLAGS
  import pocketflow_amd.nets.resnet_at_ilsvrc12  # noqa: F401
  from pocketflow_amd.learners.uniform_quantization import learner as uq
  from pocketflow_amd.learners.nonuniform_quantization import learner as nuq
  from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw
  for r in host['setup_bnds_decay_rates']:
    monkeypatch.setattr(mgw, 'size', classmethod(lambda cls, n=r['mgw_size']: n))
    FLAGS.batch_size, FLAGS.enbl_multi_gpu, FLAGS.nb_smpls_train = r['batch_size'], r['enbl_multi_gpu'], r['nb_smpls_train']
    FLAGS.lrn_rate_init, FLAGS.batch_size_norm, FLAGS.enbl_warm_start = 1e-1, 256.0, r['enbl_warm_start']
    FLAGS.uql_quant_epochs = FLAGS.nuql_quant_epochs = 60
    fn = uq.setup_bnds_decay_rates if r['learner'] == 'uq' else nuq.setup_bnds_decay_rates
    init_lr, bnds, decay, steps = fn(r['model'], r['dataset'])
    assert (init_lr, list(bnds), list(decay), steps) == (r['init_lr'], r['bnds'], r['decay_rates'], r['finetune_steps']), r


def test_maskable_vars_match_reference(host):
  from types import SimpleNamespace
  from pocketflow_amd.learners.weight_sparsification.utils import get_maskable_vars
  m = host['get_maskable_vars']
  vs = [SimpleNamespace(name=n) for n in m['names']]
  assert [v.name for v in get_maskable_vars(vs)] == m['maskable']


def test_flags_contract():
  from pocketflow_amd.flags import FLAGS, flags
  flags.DEFINE_integer('t_int', 3, '')
  flags.DEFINE_boolean('t_bool', False, '')
  flags.DEFINE_string('t_str', None, '')
  rest = FLAGS.parse(['--t_int', '7', '--t_bool', '--t_str=None', 'positional'])
  assert (FLAGS.t_int, FLAGS.t_bool, FLAGS.t_str, rest) == (7, True, None, ['positional'])
  FLAGS.parse(['--not_bool'])
  assert FLAGS.t_bool is False
  with pytest.raises(ValueError, match='Unknown command line flag'):
    FLAGS.parse(['--no_such_flag', '1'])
  with pytest.raises(AttributeError):
    _ = FLAGS.no_such_flag


def test_checkpoint_roundtrip(tmp_path):
  from pocketflow_amd.utils import checkpoint
  vals = {'model/a/kernel': np.arange(6, dtype=np.float32).reshape(1, 1, 2, 3), 'model/b': np.ones(2, np.float32)}
  assert checkpoint.latest_checkpoint(str(tmp_path)) is None
  p = checkpoint.save(vals, str(tmp_path / 'model.ckpt'), 12)
  assert p.endswith('model.ckpt-12') and checkpoint.latest_checkpoint(str(tmp_path)) == p
  back = checkpoint.load(p)
  assert sorted(back) == sorted(vals) and all(np.array_equal(back[k], vals[k]) for k in vals)
