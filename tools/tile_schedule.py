"""Tile quantisation of the persistent implicit-GEMM launches of one ResNet-50 step (B = 256 per GPU): a launch of T tiles on S resident
workgroup slots is as long as its busiest CU's tile count where the average count would do -- the CUs that finished early idle until the slowest
workgroup is done, and ONE in-order queue cannot start the next kernel on them (the second-queue experiment).
Dispatch rules as in pf_igemm.hip (ig_pick / ig_grid) and pf_conv_stream.hip (pf_conv_stream_plan).  No GPU.

    python tools/tile_schedule.py [batch]
    python tools/tile_schedule.py [batch] pieces      the LDS-DMA piece model of DESIGN.md section 9 per launch, current tile vs 256 x 256
"""
import math
import sys

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
CUS = 256
# bottleneck stages of ResNet-v2-50: (spatial size of the block OUTPUT, bottleneck width C, blocks)
STAGES = [(56, 64, 3), (28, 128, 4), (14, 256, 6), (7, 512, 3)]


def stream_ok(M, N, K):
  if K % 64 or K > 512 or N % 64 or N > 2048 or M < 4096:
    return False
  nsplit = 1
  while (N // nsplit) * K > 32768 or N // nsplit > 256:
    nsplit *= 2
  nw = N // nsplit
  return not (nsplit > 2 or (256 % (8 * nsplit)) or nw not in (64, 128, 256) or (nsplit >= 2 and N < 4 * K))


def igemm(M, N, pro, force=None):
  if force is not None:
    bm, bn, slots = force
  elif pro:
    bm, bn, slots = (128, 256, 256) if N % 256 == 0 else (256, 128, 256)
  else:
    bm, bn, slots = 128, (128 if N % 128 == 0 else 64), 512
  tm, tn = -(-M // bm), -(-N // bn)
  G = max(8, (slots // tn) // 8 * 8)
  G = min(G, -(-tm // 8) * 8)
  # tiles per CU: workgroup b = (xcd, L) owns row tiles g, g + G, ... of column L % tn with g = xcd + 8 * (L / tn); the dispatcher is
  # assumed to fill the chip breadth-first (workgroup b on CU b % 256), and two workgroups of a CU share its matrix pipe: what
  # bounds the launch is the CU with the most tiles
  per_cu = [0] * CUS
  for b in range(G * tn):
    g = (b & 7) + 8 * ((b >> 3) // tn)
    if g < tm:
      per_cu[b % CUS] += -(-(tm - g) // G)
  return '%dx%d' % (bm, bn), tm * tn, G * tn, sum(per_cu) / float(CUS), max(per_cu)


rows = []
prev_hw = 56
for si, (hw, C, blocks) in enumerate(STAGES):
  for b in range(blocks):
    first = b == 0
    in_hw = prev_hw if first else hw                      # the first block of a stage strides in its 3x3
    cin = (64 if si == 0 else STAGES[si - 1][1] * 4) if first else C * 4
    M_in, M = B * in_hw * in_hw, B * hw * hw
    for name, m, n, k, pro, taps in (('conv1 1x1', M_in, C, cin, True, 1), ('conv2 3x3 fwd', M, C, C * 9, False, 9),
                                     ('conv3 1x1 (+res)', M, C * 4, C, True, 1), ('conv2 3x3 bwd-data', M_in, C, C * 9, False, 9),
                                     ('conv3 1x1 bwd-data', M, C, C * 4, False, 1)):
      if 'bwd-data' in name and taps == 9 and first and si > 0:
        continue                                          # stride-2 backward-data: MIOpen
      if taps == 1 and stream_ok(m, n, k):
        continue                                          # resident kernel: per-wavefront strips, no tile schedule
      if taps == 1 and not pro and k < 512:
        continue                                          # tiled kernel (pf_conv.hip)
      tile, T, S, avg, rounds = igemm(m, n, pro)
      rows.append(('s%d b%d %s' % (si + 1, b, name), m, n, k, tile, T, S, avg, rounds, 2 if ('fwd' in name or pro) else 1))
  prev_hw = hw

print('%-30s %8s %5s %5s  %-8s %6s %6s %9s %9s %7s' % ('launch (x networks)', 'M', 'N', 'K', 'tile', 'tiles', 'wgs', 'tiles/CU', 'max / CU', 'idle %'))
seen = {}
tot_w = tot_i = 0.0
for name, m, n, k, tile, T, S, avg, rounds, nets in rows:
  key = (name.split(' ', 2)[2], m, n, k)
  flops = 2.0 * m * n * k * nets
  idle = 1.0 - avg / rounds
  tot_w += flops
  tot_i += flops * idle / (1.0 - idle)
  seen.setdefault(key, [0, tile, T, S, avg, rounds, idle, nets])[0] += 1
for (nm, m, n, k), (cnt, tile, T, S, avg, rounds, idle, nets) in seen.items():
  print('%-30s %8d %5d %5d  %-8s %6d %6d %9.2f %9d %6.1f%%' % ('%s x%d' % (nm, cnt * nets), m, n, k, tile, T, S, avg, rounds, idle * 100))
print('\nflop-weighted: %.1f %% of the time of these launches is CUs waiting for the busiest one (B = %d)' % (tot_i / (tot_w + tot_i) * 100, B))


# ---- the piece model: a CU lands about one 1-KiB LDS-DMA piece per PIECE_CYC cycles whatever issues it; a k-step
# of a (bm x bn) tile needs (bm + bn) / 8 pieces and bm * bn * 64 / 8192 MFMAs of 16 cycles on 4 SIMDs.  Predicted main-loop time of a
# launch = busiest CU's tiles x k-steps x max(pieces x PIECE_CYC, MFMA cycles).  Calibrated on one number (the 28x28 3x3 loop: 68 us).
if len(sys.argv) > 2 and sys.argv[2] == 'pieces':
  PIECE_CYC, GHZ = 34.0, 2.3

  def model(m, n, k, tile, max_per_cu):
    bm, bn = (int(v) for v in tile.split('x'))
    ksteps = k // 64
    dma = (bm + bn) / 8.0 * PIECE_CYC
    mfma = bm * bn * 64 / 8192.0 * 16 / 4
    return max_per_cu * ksteps * max(dma, mfma) / (GHZ * 1e3), dma, mfma

  print('\n%-30s %8s %5s %5s | %-8s %8s %9s | %-8s %8s %9s | %s' % ('launch (x networks)', 'M', 'N', 'K', 'tile', 'max / CU', 'model us', 'tile', 'max / CU',
                                                                 'model us', 'pieces : MFMA cycles per k-step'))
  t_cur = t_alt = 0.0
  for (nm, m, n, k), (cnt, tile, T, S, avg, rounds, idle, nets) in seen.items():
    pro = 'bwd' not in nm and '3x3' not in nm
    us, dma, mfma = model(m, n, k, tile, rounds)
    alt = ('-', 0, us)
    if n % 256 == 0:
      t2, T2, S2, avg2, rounds2 = igemm(m, n, pro, force=(256, 256, 256))
      us2, dma2, mfma2 = model(m, n, k, t2, rounds2)
      alt = (t2, rounds2, us2)
    t_cur += us * cnt * nets
    t_alt += min(us, alt[2]) * cnt * nets
    print('%-30s %8d %5d %5d | %-8s %8d %9.1f | %-8s %8s %9.1f | %4.0f : %4.0f' % ('%s x%d' % (nm, cnt * nets), m, n, k, tile, rounds, us, alt[0],
                                                                          alt[1] if alt[1] else '-', alt[2], dma, mfma))
  print('\nmain loops of these launches per step, model: %.2f ms with the current tiles, %.2f ms taking 256 x 256 wherever the model prefers it'
        % (t_cur / 1e3, t_alt / 1e3))
