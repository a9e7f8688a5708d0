"""Static guards for the rules this repository is built under: the product never touches the oracle, nothing that
runs on the GPU box reads /root/reference, every kernel source is part of the build, the header and the ctypes
binding agree.  No GPU."""
import ast
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py(pattern):
  return sorted(glob.glob(os.path.join(ROOT, pattern), recursive=True))


def _imports(path):
  tree = ast.parse(open(path).read())
  for node in ast.walk(tree):
    if isinstance(node, ast.Import):
      for a in node.names:
        yield a.name
    elif isinstance(node, ast.ImportFrom) and node.module:
      yield node.module


def test_product_never_imports_the_oracle_or_the_tests():
  for path in _py('pocketflow_amd/**/*.py'):
    for mod in _imports(path):
      top = mod.split('.')[0]
      assert top not in ('oracle', 'tests', 'fake_hip'), '%s imports %s' % (os.path.relpath(path, ROOT), mod)
    assert not re.search(r'''(import_module|__import__)\(\s*['"](oracle|tests|fake_hip)''', open(path).read()), os.path.relpath(path, ROOT)


def test_only_allowed_callers_use_the_oracle():
  allowed = {'bench.py', '__graft_entry__.py'}
  for path in _py('*.py') + _py('tools/**/*.py') + _py('scripts/**/*.py'):
    rel = os.path.relpath(path, ROOT)
    uses = any(m.split('.')[0] == 'oracle' for m in _imports(path))
    if uses:
      assert rel in allowed or rel.startswith('tools/gpu/'), rel      # tools/gpu: diagnostics, not shipped paths
  # bench.py may only use it in its cpu_baseline leg
  src = open(os.path.join(ROOT, 'bench.py')).read()
  for m in re.finditer(r'oracle', src):
    ctx = src[max(0, m.start() - 1500):m.start()]
    assert 'cpu_baseline' in ctx or 'cpu baseline' in ctx.lower() or 'time_cpu_baseline' in src[m.start():m.start() + 200], m.start()


def test_nothing_on_the_gpu_box_reads_the_reference_tree():
  for path in _py('tests/*.py') + _py('pocketflow_amd/**/*.py') + [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')]:
    src = open(path).read()
    code = '\n'.join(l for l in src.split('\n'))
    for m in re.finditer(r'/root/reference', code):
      line = code[code.rfind('\n', 0, m.start()) + 1:code.find('\n', m.start())]
      # only prose (docstrings / comments citing reference files) may mention the tree
      assert 'open(' not in line and 'sys.path' not in line and 'os.path.join' not in line, (os.path.relpath(path, ROOT), line)
  # the fixture generators (build container only) are the one place that opens it
  gens = [os.path.basename(p) for p in _py('tests/golden/*.py')]
  assert sorted(gens) == ['make_reference_cp_features_golden.py', 'make_reference_cpg_golden.py', 'make_reference_flags.py', 'make_reference_golden.py',
                          'make_reference_image_golden.py', 'make_reference_rl_golden.py']


def test_every_kernel_source_is_built_and_every_symbol_is_bound():
  build = open(os.path.join(ROOT, 'pocketflow_amd/csrc/build.sh')).read()
  for path in _py('pocketflow_amd/csrc/*.hip'):
    assert os.path.basename(path)[:-4] in build, path
  header = open(os.path.join(ROOT, 'include/pocketflow_hip.h')).read()
  declared = set(re.findall(r'\b(pf_[a-z0-9_]+)\s*\(', header))
  from pocketflow_amd import hip
  assert declared == set(hip.SYMBOLS), declared ^ set(hip.SYMBOLS)
  # struct layouts mirrored in NumPy dtypes
  import ctypes

  class _Desc(ctypes.Structure):
    _fields_ = [('offset', ctypes.c_int64), ('h', ctypes.c_int32), ('w', ctypes.c_int32), ('scale_y', ctypes.c_float),
                ('scale_x', ctypes.c_float), ('off_y', ctypes.c_int32), ('off_x', ctypes.c_int32), ('flip', ctypes.c_int32),
                ('reserved', ctypes.c_int32)]
  assert ctypes.sizeof(_Desc) == hip.IMAGE_DESC_DTYPE.itemsize == 40
  for f, _t in _Desc._fields_:
    assert getattr(_Desc, f).offset == hip.IMAGE_DESC_DTYPE.fields[f][1], f


def test_every_reference_flag_on_the_path_exists_with_the_same_default():
  """The command lines of the reference stay valid: every tf.app.flags definition of the reference's learners / nets /
  datasets / utils / rl_agents on the path (tests/golden/reference_flags.json) is defined here under the same name, and
  flags the reference defines once have the same default."""
  import json
  import pocketflow_amd.nets.run_utils  # noqa: F401
  import pocketflow_amd.learners.learner_utils  # noqa: F401
  for mod in ('abstract_learner', 'distillation_helper', 'full_precision.learner', 'uniform_quantization.learner',
              'nonuniform_quantization.learner', 'weight_sparsification.learner', 'channel_pruning.learner', 'channel_pruning_gpu.learner'):
    __import__('pocketflow_amd.learners.' + mod)
  for mod in ('resnet_at_ilsvrc12', 'mobilenet_at_ilsvrc12', 'lenet_at_cifar10', 'resnet_at_cifar10'):
    __import__('pocketflow_amd.nets.' + mod)
  from pocketflow_amd.flags import FLAGS
  FLAGS.reset()
  ours = FLAGS.flag_values_dict()
  ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_flags.json')))
  missing = sorted(k for k in ref if k not in ours)
  assert not missing, missing
  for name, defs in ref.items():
    defaults = {json.dumps(d['default']) for d in defs}
    if len(defaults) == 1 and not isinstance(defs[0]['default'], str) or (len(defaults) == 1 and defs[0]['kind'] == 'string'):
      want = defs[0]['default']
      got = ours[name]
      if isinstance(want, float) or isinstance(got, float):
        assert got is not None and abs(float(got) - float(want)) <= 1e-12 * max(1.0, abs(float(want))), (name, got, want)
      else:
        assert got == want, (name, got, want)


def test_gpu_tools_parse_and_name_files_that_exist():
  """tools/gpu/ is what a GPU session runs first (tools/gpu/round_evidence.sh): a syntax error or a renamed script there costs
  GPU minutes.  Every Python tool must parse, every shell script must pass `bash -n`, and every `tools/...` path a script names must
  exist."""
  import subprocess
  tools = os.path.join(ROOT, 'tools')
  for path in glob.glob(os.path.join(tools, '**', '*.py'), recursive=True):
    ast.parse(open(path).read(), filename=path)
  for path in glob.glob(os.path.join(tools, 'gpu', '*.sh')):
    r = subprocess.run(['bash', '-n', path], capture_output=True, text=True)
    assert r.returncode == 0, (path, r.stderr)
    for ref in re.findall(r'\btools/[\w/]+\.(?:py|sh)\b', open(path).read()):
      assert os.path.exists(os.path.join(ROOT, ref)), (path, ref)


def test_contraction_kernels_compile_without_spills(tmp_path):
  """The contraction kernels with the sched_group_barrier pipelines (the product since round 4, profiles/r04_first_call_ab.txt): none
  of the kernels the step dispatches to may spill vector registers or need more than 256 of them (two wavefronts per SIMD) -- a
  spill in a main loop costs more than any of the scheduling changes gained and shows in no test."""
  import subprocess
  hipcc = '/opt/rocm/bin/hipcc'
  if not os.path.exists(hipcc):
    import pytest
    pytest.skip('no hipcc')
  flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt', '-S',
           '--cuda-device-only']
  procs = {}
  for f in ('pf_igemm', 'pf_conv_stream', 'pf_wrw'):
    out = str(tmp_path / (f + '.s'))
    procs[f] = (out, subprocess.Popen([hipcc] + flags + [os.path.join(ROOT, 'pocketflow_amd', 'csrc', f + '.hip'), '-o', out],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
  dispatched = re.compile(r'k_igemmILi128ELi256ELi2ELi4ELi3ELi2E|k_igemmILi256ELi128ELi4ELi2ELi3ELi2E|k_igemmILi128ELi128ELi2ELi2ELi2ELi[01]E|'
                          r'k_igemmILi128ELi64ELi2ELi2ELi2ELi[01]E|k_conv1x1_streamILi(64|256)E|k_wrw2I')
  for f, (out, p) in procs.items():
    _, err = p.communicate(timeout=600)
    assert p.returncode == 0, (f, err[-2000:])
    text = open(out).read()
    for m in re.finditer(r'\.name:\s+(\S+)(.*?)\.wavefront_size', text, re.S):
      name, blk = m.group(1), m.group(2)
      if not dispatched.search(name):
        continue
      vg = int(re.search(r'\.vgpr_count:\s+(\d+)', blk).group(1))
      sp = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', blk).group(1))
      assert sp == 0 and vg <= 256, (name, vg, sp)
