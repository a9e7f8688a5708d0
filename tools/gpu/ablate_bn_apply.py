"""Upper bound of what folding k_bn_bwd_apply into its consumers can give, measured instead of argued:
bench.py's default command with the apply pass of every FOLDABLE BatchNorm dropped -- bn2 / bn3 of a bottleneck block, whose dx has
exactly two consumers (the producing convolution's backward-filter and backward-data); the block's bn1 (4C channels, shortcut gradient
added in the same pass) feeds three consumers and stays.  The dropped pass hands dq on as if it were dx, so the RESULTS ARE GARBAGE BY
DESIGN and the line is no benchmark: the difference to the unpatched command in the same box is the ceiling of ANY fold (a real fold
still has to read dq and x inside a consumer: it keeps at least two thirds of the dropped traffic).
  python tools/gpu/ablate_bn_apply.py [mode] [bench.py arguments]     mode: none | foldable | all"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'foldable'
sys.argv = ['bench.py'] + [a for a in sys.argv[1:] if a != mode or a.startswith('-')]
import bench
from pocketflow_amd import hip
dropped = {'n': 0, 'kept': 0}
orig = hip.bn_bwd_apply


def patched(dq, x, dx, rows, C, scale_shift, mean_invstd, dgamma, dbeta, act, addend=None):
  # bn2 / bn3 of the bottleneck blocks by shape (B = 256): (H, C) of the tensors behind conv1 / conv2.  bn1 of a projection block also
  # comes without an addend but has three consumers; the one shape both kinds share -- (56, 64): the pool output's bn1 -- is dropped
  # with the six bn2 / bn3 launches of that shape (one launch of ~50 us over-counted: this is a CEILING)
  foldable = addend is None and (rows // 256, C) in ((56 * 56, 64), (56 * 56, 128), (28 * 28, 128), (28 * 28, 256), (14 * 14, 256),
                                                      (14 * 14, 512), (7 * 7, 512))
  if mode == 'all' or (mode == 'foldable' and foldable):
    dropped['n'] += 1
    return None                      # dx stays uninitialised memory: garbage by design
  dropped['kept'] += 1
  return orig(dq, x, dx, rows, C, scale_shift, mean_invstd, dgamma, dbeta, act, addend)


if mode != 'none':
  hip.bn_bwd_apply = patched
bench.main()
sys.stderr.write('ablate_bn_apply: mode %s, apply launches dropped %d, kept %d (over all steps issued launch by launch or recorded)\n' % (mode, dropped['n'], dropped['kept']))
