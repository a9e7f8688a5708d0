"""What hipcc emits for a kernel's contraction loop, as a skeleton: LDS reads, waits, barriers and runs of MFMAs between the first and
the last MFMA of the kernel, plus register / spill counts and how many v_max_f32 / v_min_f32 the whole kernel holds.  No GPU needed.

    python tools/isa_skeleton.py pf_igemm.hip '_Z7k_igemmILi128ELi256ELi2ELi4ELi3ELi2EEv6IgArgs' [-DPF_SOME_EXPERIMENT ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt',
         '-S', '--cuda-device-only']


def compile_asm(src, defines):
  out = tempfile.NamedTemporaryFile(suffix='.s', delete=False).name
  subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + list(defines) + [os.path.join(ROOT, 'pocketflow_amd', 'csrc', src), '-o', out],
                 check=True, capture_output=True)
  text = open(out).read()
  os.unlink(out)
  return text


def kernel_body(text, name):
  lines = text.splitlines()
  start = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
  end = next(i for i in range(start, len(lines)) if '.end_amdhsa_kernel' in lines[i] or lines[i].startswith('.Lfunc_end'))
  return lines[start:end]


def meta(text, name):
  m = re.search(r'\.name:\s+%s\b(.*?)\.wavefront_size' % re.escape(name), text, re.S)
  blk = m.group(1) if m else ''
  get = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)) if re.search(r'\.%s:\s+(\d+)' % k, blk) else None
  return get('vgpr_count'), get('vgpr_spill_count'), get('sgpr_spill_count')


def skeleton(body, limit=70):
  idx = [i for i, l in enumerate(body) if 'v_mfma' in l]
  if not idx:
    return ['(no MFMA)']
  out, run = [], 0
  for l in body[max(0, idx[0] - 14):idx[-1] + 2]:
    t = l.strip().split(';')[0].strip()
    op = t.split(' ')[0].split('\t')[0] if t else ''
    if op.startswith('v_mfma'):
      run += 1
      continue
    if op.startswith(('ds_read', 'ds_write', 's_waitcnt', 's_barrier', 'buffer_load', 'global_load')) or 'ASMSTART' in l:
      if run:
        out.append('      mfma x%d' % run)
        run = 0
      out.append('  ' + ('(inline asm)' if 'ASMSTART' in l else re.sub(r'\s+', ' ', t)[:60]))
  if run:
    out.append('      mfma x%d' % run)
  return out[:limit] + (['  ... (%d more lines)' % (len(out) - limit)] if len(out) > limit else [])


def main(argv):
  src, name, defines = argv[1], argv[2], argv[3:]
  text = compile_asm(src, defines)
  body = kernel_body(text, name)
  v, sp, ssp = meta(text, name)
  print('%s  %s  %s' % (src, name, ' '.join(defines) or '(product flags)'))
  print('  registers %s, spilled vector registers %s, scalar spills %s | v_max_f32 %d, v_min_f32 %d, ds_read %d, instructions %d' % (
      v, sp, ssp, sum('v_max_f32' in l for l in body), sum('v_min_f32' in l for l in body), sum('ds_read' in l for l in body),
      sum(bool(re.match(r'\s+[vs]_|\s+ds_|\s+buffer_|\s+global_', l)) for l in body)))
  for l in skeleton(body):
    print(l)


if __name__ == '__main__':
  main(sys.argv)
