#!/usr/bin/env python
"""Compare two device-assembly snapshots (tools/isa_snapshot.sh) kernel by kernel.

    tools/isa_snapshot.sh /tmp/isa_before;  <refactor>;  tools/isa_snapshot.sh /tmp/isa_after
    python tools/isa_diff.py /tmp/isa_before /tmp/isa_after

For every function symbol the instruction stream (labels renumbered per function, comments and directives dropped) and the
kernel descriptor's resource lines (.amdhsa_*: registers, LDS, scratch) are compared.  Prints which kernels are identical,
changed, added or removed; exit status 1 when a kernel present on both sides differs.  This is how a clean-up of kernel SOURCE
(a deleted template mode, a removed tuning switch) is shown to leave the shipped code untouched when no GPU is at hand.
"""
import os
import re
import sys

LABEL = re.compile(r'\.L(BB|func_begin|func_end|tmp|JTI)\d+(_\d+)?')


def functions(path):
  """{symbol: (instructions, descriptor lines)} of one .s file."""
  out, cur, body = {}, None, []
  desc, in_desc, desc_of = {}, None, None
  meta = False
  for raw in open(path, errors='replace'):
    line = raw.split(';', 1)[0].rstrip()
    s = line.strip()
    if s.startswith('.amdgpu_metadata') or s.startswith('.end_amdgpu_metadata'):
      meta = s.startswith('.amdgpu_metadata')    # the YAML block lists every kernel of the file: not per-kernel code
      cur = None
      continue
    if not s or meta:
      continue
    m = re.match(r'^(\w[\w$.]*):$', s)
    if m and not s.startswith('.L'):
      if m.group(1).endswith('.kd') or cur is not None and s.startswith('.L'):
        continue
      cur, body = m.group(1), []
      out[cur] = body
      continue
    if s.startswith('.amdhsa_kernel '):
      desc_of = s.split()[1]
      in_desc = desc.setdefault(desc_of, [])
      continue
    if s == '.end_amdhsa_kernel':
      in_desc = None
      continue
    if in_desc is not None:
      in_desc.append(s)
      continue
    if cur is None:
      continue
    if s.startswith('.Lfunc_end'):
      cur = None
      continue
    if s.startswith('.') and not s.startswith('.L'):
      continue                                   # directives (.p2align, .section ...)
    body.append(s)
  res = {}
  for name, ins in out.items():
    labels = {}

    def renum(mm):
      return labels.setdefault(mm.group(0), '.L%d' % len(labels))
    if not name.startswith('__hip_cuid_'):       # a per-compilation identifier derived from the source text, no code
      res[name] = ([LABEL.sub(renum, i) for i in ins], desc.get(name, []))
  return res


def main():
  a_dir, b_dir = sys.argv[1:3]
  rc = 0
  names = sorted(set(os.listdir(a_dir)) | set(os.listdir(b_dir)))
  for f in names:
    if not f.endswith('.s'):
      continue
    pa, pb = os.path.join(a_dir, f), os.path.join(b_dir, f)
    if not os.path.exists(pa) or not os.path.exists(pb):
      print('%-22s %s' % (f, 'only in ' + (a_dir if os.path.exists(pa) else b_dir)))
      continue
    A, B = functions(pa), functions(pb)
    same = [k for k in A if k in B and A[k] == B[k]]
    diff = [k for k in A if k in B and A[k] != B[k]]
    gone = [k for k in A if k not in B]
    new = [k for k in B if k not in A]
    n_ins = sum(len(A[k][0]) for k in same)
    print('%-22s identical %3d (%7d instructions)  changed %d  removed %d  added %d' % (f, len(same), n_ins, len(diff), len(gone), len(new)))
    for k in diff:
      ia, ib = A[k][0], B[k][0]
      first = next((i for i, (x, y) in enumerate(zip(ia, ib)) if x != y), min(len(ia), len(ib)))
      print('    CHANGED %s: %d -> %d instructions, first difference at %d; descriptor %s' % (
          k, len(ia), len(ib), first, 'same' if A[k][1] == B[k][1] else 'differs'))
      rc = 1
    for k in gone:
      print('    removed %s' % k)
    for k in new:
      print('    added   %s' % k)
  return rc


if __name__ == '__main__':
  sys.exit(main())
