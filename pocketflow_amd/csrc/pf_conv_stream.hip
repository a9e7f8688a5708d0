// K12 fused with K13/K4 for the HBM-bound 1x1 convolutions (ResNet-50 stages 1-2, MobileNet pointwise layers whose
// kernel fits the LDS): the same operation as pf_conv1x1_fwd in pf_conv.hip
//
//     Y = conv1x1( Q(x) ) [+ R],   Q(x) = fake_quant(act(scale*x + shift)),   partial = per-channel statistics of Y
//     (reference chain: utils/external/resnet_model.py:257-314 + learners/uniform_quantization/utils.py:51-79)
//
// built for a different regime.  With K*N <= 64 Ki the whole (slice of the) kernel stays RESIDENT in LDS for the
// lifetime of a persistent workgroup, so nothing is shared between wavefronts while they run:
//   * every wavefront owns strips of 16 / 32 pixel rows and all NW output channels of its workgroup: it loads its
//     input rows straight into MFMA operand registers (lane (p, q) = 16 bytes of pixel p at channel group q: a
//     fragment-shaped, per-lane addressed load -- strided / remapped rows cost nothing), applies the producer's
//     BN + ReLU + fake-quant on those registers, multiplies against weight fragments read from LDS, and transposes its
//     own output tile through a wave-private LDS buffer for 16-byte row stores, the residual add and the statistics;
//   * there is NO workgroup barrier in the main loop (one after the weights are staged, one for the final statistics):
//     a wavefront stalls only on its own loads, and the 8 wavefronts of a CU drift apart so that one's prologue VALU
//     work overlaps another's MFMAs and a third's memory wait -- the 2-barriers-per-k-step structure of pf_conv.hip
//     serialised those (measured there: +40 % prologue, +25 % residual, +10 % statistics, all additive);
//   * input chunks are prefetched D deep in registers, the residual / BN-input vectors of the NEXT strip are
//     re-requested into the registers the epilogue has just consumed.
// Layers whose kernel is 128 KiB (128->512, 512->128) are split over two workgroups per row panel (column halves)
// that sit on the same XCD (blockIdx % 8), so the second read of the panel is served by that XCD's L2.
//
// MFMA orientation as in pf_conv.hip: weights = operand A, pixels = operand B, so a lane's 4 accumulator values
// are 4 consecutive output channels of one pixel.
#include "pf_conv_common.h"
#include <stdlib.h>

#define ST_THREADS 512
#define ST_WAVES 8
#define ST_GRID 256                 // one persistent workgroup per CU

// MAP: strided / remapped rows (projection shortcuts); the stride-1 instantiations carry no index-division code at all
// (it is inlined at every load and store site: without the split the loop body outgrows the instruction cache)
// AFF: the output affine of the row pass (ConvArgs.oss; forward launches of an inference-mode network only) -- a template
// parameter, not a run-time branch: the branch cost the training kernels 8-16 registers (k_conv1x1_stream<256, true> spilled)
template <int NW, bool PRO, bool BWD, bool MAP, bool AFF = false>
__global__ __launch_bounds__(ST_THREADS) void k_conv1x1_stream(const ConvArgs a, const int nsplit) {
  constexpr int JM = (NW == 256) ? 1 : 2;           // 16-pixel blocks per strip
  constexpr int RS = 16 * JM;                       // pixel rows per strip
  constexpr int NI = NW / 16;                       // 16-channel blocks
  // NW = 128: two chunks in flight (three spill 3-21 registers; measured round 4, 56x56 256->128: 147 -> 140 us)
  constexpr int D = (NW == 128) ? 2 : 4;             // input chunks (RS rows x 64 channels) in flight per wavefront (8-16 KiB).
  // Deep on purpose: gfx950 counts loads and stores on ONE counter (vmcnt) and they may retire out of order with
  // respect to each other, so the compiler drains the counter whenever a load result is needed while a store is
  // pending -- once per strip here.  What keeps HBM busy across that drain is the amount each wavefront has in flight.
  constexpr int CS_LD = NW + 8;
  constexpr int VPR = NW / 8;                       // 16-byte vectors per output row
  constexpr int RPP = 64 / VPR;                     // rows per epilogue pass of one wavefront
  constexpr int NP = RS / RPP;                      // passes
  constexpr int NR = 2 * JM;                        // 16-byte registers per chunk

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K, KC = K >> 6, KV = K >> 3;
  bf16_t* Wl = reinterpret_cast<bf16_t*>(smem);                          // [NW][K], 16-byte groups XOR-swizzled
  float* aux = reinterpret_cast<float*>(smem + (size_t)NW * K * 2);      // PRO: scale | shift [2][K];  BWD: [4][NW]
  const int aux_fl = PRO ? 2 * K : (BWD ? 4 * NW : 0);
  bf16_t* Cs_all = reinterpret_cast<bf16_t*>(smem + (size_t)NW * K * 2 + (size_t)aux_fl * 4);
  float* red = reinterpret_cast<float*>(Cs_all);                         // final statistics reduction [4][8][NW]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, q = lane >> 4;
  bf16_t* Cs = Cs_all + wave * (RS * CS_LD);

  // workgroup -> (row-panel set g, column slice): the nsplit slices of one panel set share an XCD
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int slice = idx % nsplit, g = (idx / nsplit) * 8 + xcd;
  const int nsets = gridDim.x / nsplit;
  const int n0 = slice * NW;
  const int sstep = nsets * ST_WAVES;                                    // strip stride of one wavefront
  const int NS = (a.M + RS - 1) / RS;
  const int s_first = g * ST_WAVES + wave;
  const int cnt = (s_first < NS) ? (NS - s_first + sstep - 1) / sstep : 0;
  const int T = cnt * KC;

  Pro pro;
  pro_init(pro, a.act_lo, a.act_hi, a.kq, PRO ? a.slot : nullptr);
  // ---- stage the kernel slice (and the per-channel vectors) once ------------------------------------------------
  {
    const int swz_mask = (KV >= 16) ? 15 : 7;
    for (int v = tid; v < NW * KV; v += ST_THREADS) {
      const int n = v / KV, c = v - n * KV;
      uint4 w = make_uint4(0, 0, 0, 0);
      if (n0 + n < a.N) w = *reinterpret_cast<const uint4*>(a.W + (int64_t)(n0 + n) * K + c * 8);
      *reinterpret_cast<uint4*>(Wl + ((int64_t)n * KV + (c ^ (n & swz_mask))) * 8) = w;
    }
    if (PRO) for (int i = tid; i < 2 * K; i += ST_THREADS)   // FOLDED scale | shift (pf_conv_common.h)
      aux[i] = (i < K) ? pro_fold_scale(pro, a.ss[i]) : pro_fold_shift(pro, a.ss[i]);
    if (BWD) for (int i = tid; i < 4 * NW; i += ST_THREADS) {
      const int qq = i / NW, c = n0 + (i - qq * NW);
      float v = 0.f;
      if (c < a.N) v = (qq < 2) ? a.bss[qq * a.N + c] : a.bmi[(qq - 2) * a.N + c];
      aux[i] = v;
    }
  }
  __syncthreads();

  const int wswz = (KV >= 16) ? l15 : (l15 & 7);                          // this lane's weight-row swizzle
  const bf16_t* __restrict__ side = BWD ? a.bx : nullptr;                // second [M][N] operand of the row pass
  const bool has_r = !BWD && a.R != nullptr;                             // residual: added on the fp32 accumulators
  const int wvec = lane % VPR, wrow = lane / VPR;

  f32x4 acc[NI][JM];
  uint4 ring[D][NR];
  uint4 rres[NP];
  uint2 rr[NI][JM];                                                      // residual of the NEXT strip, accumulator layout
  float st_s[8], st_q[8], st_mn[8], st_mx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; st_mn[j] = INFINITY; st_mx[j] = -INFINITY; }

  // load-side cursor (chunk t -> strip, 64-channel step)
  int ld_s = s_first, ld_kc = 0;
  auto issue = [&](uint4 (&r)[NR]) {
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      int m = ld_s * RS + j * 16 + l15;
      m = (m < a.M) ? m : (a.M - 1);                                     // tail rows: any valid address (never stored)
      const int64_t row = (!MAP || a.ymap) ? (int64_t)m : map_row(a, m);
      const bf16_t* p = a.X + row * K + ld_kc * 64 + q * 8;
      r[j * 2 + 0] = *reinterpret_cast<const uint4*>(p);
      r[j * 2 + 1] = *reinterpret_cast<const uint4*>(p + 32);
    }
    if (++ld_kc == KC) { ld_kc = 0; ld_s += sstep; }
  };
  auto issue_side = [&](int s, int p) {
    const int m = s * RS + p * RPP + wrow, n = n0 + wvec * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < a.M && n < a.N) {
      const int64_t orow = (MAP && a.ymap) ? map_row(a, m) : (int64_t)m;
      v = *reinterpret_cast<const uint4*>(side + orow * a.N + n);
    }
    rres[p] = v;
  };

  // residual of strip s in the accumulator layout (this lane: pixel (j, l15), four channels of block i = 8 bytes)
  auto issue_res = [&](int s) {
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const int m = s * RS + j * 16 + l15;
      const int64_t orow = (m < a.M) ? ((MAP && a.ymap) ? map_row(a, m) : (int64_t)m) : 0;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int n = n0 + i * 16 + q * 4;
        uint2 v = make_uint2(0, 0);
        if (m < a.M && n < a.N) v = *reinterpret_cast<const uint2*>(a.R + orow * a.N + n);
        rr[i][j] = v;
      }
    }
  };
#pragma clang loop unroll(full)
  for (int d = 0; d < D; ++d)
    if (d < T) issue(ring[d]);
  if (has_r && cnt > 0) issue_res(s_first);
  if (side != nullptr && cnt > 0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) issue_side(s_first, p);
  }

  int cs = s_first, ckc = 0;                                             // compute-side cursor
  int slot = 0;                                                          // ring slot of chunk t (wave-uniform)
  for (int t = 0; t < T; ++t) {
    // take chunk t out of its ring slot and refill the slot at once.  The slot index is wave-uniform: a scalar branch
    // per slot keeps every register index static while the multiply / epilogue code below exists ONCE.
    uint4 v[NR];
    const bool refill = t + D < T;
#pragma clang loop unroll(full)
    for (int d = 0; d < D; ++d) {
      if (slot == d) {
#pragma unroll
        for (int i = 0; i < NR; ++i) v[i] = ring[d][i];
        if (refill) issue(ring[d]);
      }
    }
    slot = (slot + 1 == D) ? 0 : slot + 1;
    if (ckc == 0) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < JM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (PRO) {
        const int k = ckc * 64 + kk * 32 + q * 8;
        const float4 s0 = *reinterpret_cast<const float4*>(aux + k), s1 = *reinterpret_cast<const float4*>(aux + k + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(aux + K + k), h1 = *reinterpret_cast<const float4*>(aux + K + k + 4);
        pro.sc[0] = s0.x; pro.sc[1] = s0.y; pro.sc[2] = s0.z; pro.sc[3] = s0.w;
        pro.sc[4] = s1.x; pro.sc[5] = s1.y; pro.sc[6] = s1.z; pro.sc[7] = s1.w;
        pro.sh[0] = h0.x; pro.sh[1] = h0.y; pro.sh[2] = h0.z; pro.sh[3] = h0.w;
        pro.sh[4] = h1.x; pro.sh[5] = h1.y; pro.sh[6] = h1.z; pro.sh[7] = h1.w;
      }
      bf16x8 xf[JM];
#pragma unroll
      for (int j = 0; j < JM; ++j) {
        uint4 u = v[j * 2 + kk];
        if (PRO) u = (NW == 128) ? pro_apply_scalar(pro, u) : pro_apply(pro, u);
        xf[j] = *reinterpret_cast<const bf16x8*>(&u);
      }
      const int c = (ckc * 8 + kk * 4 + q) ^ wswz;
      constexpr int WB = 4;                                                // weight fragments read per batch
#pragma unroll
      for (int ib = 0; ib < NI; ib += WB) {
        bf16x8 wf[WB];
#pragma unroll
        for (int i = 0; i < WB; ++i)
          wf[i] = *reinterpret_cast<const bf16x8*>(Wl + ((int64_t)((ib + i) * 16 + l15) * KV + c) * 8);
#pragma unroll
        for (int i = 0; i < WB; ++i)
#pragma unroll
          for (int j = 0; j < JM; ++j)
            acc[ib + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[ib + i][j], 0, 0, 0);
      }
      // Experiment (tools/gpu/build_ablate.sh -> libst_sgb.so): the batches as written -- WB fragment reads, then their MFMAs
      // (with the prologue, the four reads of the folded constants come first).  Left alone, hipcc keeps two fragment registers
      // and runs read -> lgkmcnt(1) -> ONE MFMA -> read -> ... : an exposed LDS round trip per MFMA (ISA, round 3).
      // Requested: a ROLLING window -- DEPTH fragment reads ahead, then one read per group of MFMAs, so that the LDS latency of a
      // fragment is covered by the MFMAs of the DEPTH fragments before it (what -amdgpu-sched-strategy=max-ilp produces for this
      // loop; as a global flag it blows the plain igemm kernels up to 300 registers, so the pipeline is prescribed here instead).
      if (PRO) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      constexpr int DEPTH = (NW == 128) ? 2 : 4;
      __builtin_amdgcn_sched_group_barrier(0x100, DEPTH, 0);
#pragma unroll
      for (int i = 0; i < NI - DEPTH; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, JM, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, DEPTH * JM, 0);
    }
    const bool last = (ckc == KC - 1);
    const int s = cs;
    if (++ckc == KC) { ckc = 0; cs += sstep; }
    if (!last) continue;

    // ---- epilogue of one [RS][NW] tile: wave-private transposition, no workgroup barrier ----------------------
    if (has_r) {                                                         // ONE rounding to bf16 (in the staging below)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < JM; ++j) {
          acc[i][j][0] += __uint_as_float(rr[i][j].x << 16); acc[i][j][1] += __uint_as_float(rr[i][j].x & 0xFFFF0000u);
          acc[i][j][2] += __uint_as_float(rr[i][j].y << 16); acc[i][j][3] += __uint_as_float(rr[i][j].y & 0xFFFF0000u);
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < JM; ++j) {
        const uint2 w = make_uint2(pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3]));
        *reinterpret_cast<uint2*>(Cs + (j * 16 + l15) * CS_LD + i * 16 + q * 4) = w;
      }
    const bool more = (cs < NS);                                         // this wavefront has another strip
    if (has_r && more) issue_res(cs);                                    // next strip's residual travels under its main loop
    float bpr[32];                                                       // BWD: this lane's 8 channels of scale | shift | mean | invstd
    if (BWD && NW != 128) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int j = 0; j < 8; ++j) bpr[qq * 8 + j] = aux[qq * NW + wvec * 8 + j];
    } else if (BWD) {
#pragma unroll
      for (int i = 0; i < 32; ++i) bpr[i] = 0.f;
    }
    constexpr int CB = (NW == 128) ? 1 : 4;                              // scheduling experiment: rows requested CB at a time
    uint4 cv[CB];                                                        // (NW = 128 has no registers to spare: it spills as it is)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int rl = p * RPP + wrow;
      const int m = s * RS + rl, n = n0 + wvec * 8;
      if (p % CB == 0) {
#pragma unroll
        for (int pb = 0; pb < CB; ++pb)
          if (p + pb < NP) cv[pb] = *reinterpret_cast<const uint4*>(Cs + ((p + pb) * RPP + wrow) * CS_LD + wvec * 8);
      }
      uint4 c = cv[p % CB];
      const uint4 sv = rres[p];
      if (side != nullptr && more) issue_side(cs, p);                    // next strip's vector into the freed register
      if (m < a.M && n < a.N) {
        const int64_t orow = (MAP && a.ymap) ? map_row(a, m) : (int64_t)m;
        if (BWD) {
          float f[8], xv[8];
          unpack8(c, f);
          unpack8(sv, xv);
          const float* bp = (NW == 128) ? (aux + wvec * 8) : bpr;        // hoisted: hipcc re-reads the 8 vectors from LDS in every pass
          constexpr int BPS = (NW == 128) ? NW : 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float u = fmaf(bp[j], xv[j], bp[BPS + j]);
            const float dy = (u > a.b_lo && u < a.b_hi) ? f[j] : 0.f;
            st_s[j] += dy;
            st_q[j] = fmaf(dy, (xv[j] - bp[2 * BPS + j]) * bp[3 * BPS + j], st_q[j]);
          }
        } else if (a.R != nullptr || a.partial != nullptr) {
          float f[8];
          unpack8(c, f);
          if (a.partial != nullptr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              st_s[j] += f[j];
              st_q[j] = fmaf(f[j], f[j], st_q[j]);
              st_mn[j] = pf_acc_min(st_mn[j], f[j]);
              st_mx[j] = pf_acc_max(st_mx[j], f[j]);
            }
          }
        }
        if constexpr (AFF) c = out_affine8(c, a.oss, a.N, n, a.oact);
        *reinterpret_cast<uint4*>(a.Y + orow * a.N + n) = c;
      }
    }
  }

  // ---- statistics: lanes with equal column group -> wavefronts -> partial[g][stat][n0 + c], fixed order --------
  if (a.partial != nullptr) {
#pragma unroll
    for (int o = VPR; o < 64; o <<= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        st_s[j] += __shfl_xor(st_s[j], o, 64);
        st_q[j] += __shfl_xor(st_q[j], o, 64);
        st_mn[j] = fminf(st_mn[j], __shfl_xor(st_mn[j], o, 64));
        st_mx[j] = fmaxf(st_mx[j], __shfl_xor(st_mx[j], o, 64));
      }
    }
    __syncthreads();                                                     // every wavefront is done with its staging tile
    if (lane < VPR) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(0 * ST_WAVES + wave) * NW + lane * 8 + j] = st_s[j];
        red[(1 * ST_WAVES + wave) * NW + lane * 8 + j] = st_q[j];
        red[(2 * ST_WAVES + wave) * NW + lane * 8 + j] = st_mn[j];
        red[(3 * ST_WAVES + wave) * NW + lane * 8 + j] = st_mx[j];
      }
    }
    __syncthreads();
    const int nstat = BWD ? 2 : 4;
    for (int t = tid; t < nstat * NW; t += ST_THREADS) {
      const int stat = t / NW, c = t - stat * NW;
      float v = red[(stat * ST_WAVES) * NW + c];
      for (int w = 1; w < ST_WAVES; ++w) {
        const float x = red[(stat * ST_WAVES + w) * NW + c];
        v = (stat < 2) ? (v + x) : (stat == 2 ? fminf(v, x) : fmaxf(v, x));
      }
      if (n0 + c < a.N) a.partial[((int64_t)g * nstat + stat) * a.N + n0 + c] = v;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int stream_enabled() {
  return pf_tuning().conv_stream;                         // PF_CONV_STREAM=0: tuning / A-B override
}

static int stream_max_split() {
  return pf_tuning().conv_stream_maxsplit;                // PF_CONV_STREAM_MAXSPLIT: column slices a row panel may be cut into
}

// column slices (0: the stream kernel does not apply) and the slice width
int pf_conv_stream_plan(int M, int N, int K, int* nw_out) {
  if (!stream_enabled()) return 0;
  if ((K % 64) || K > 512 || (N % 64) || N > 2048) return 0;
  if (M < 4096) return 0;                                 // too few strips for 2048 persistent wavefronts
  int nsplit = 1;
  while ((int64_t)(N / nsplit) * K > 32768 || N / nsplit > 256) nsplit *= 2;
  const int nw = N / nsplit;
  if (nsplit > stream_max_split() || (ST_GRID % (8 * nsplit)) || (nw != 64 && nw != 128 && nw != 256)) return 0;
  // column slices re-read the input panel (from L2) and repeat its prologue: only worth it when the input is the
  // small operand (128 -> 512, 256 -> 1024); 512 -> 128 stays on the tiled kernel (measured: 109 vs 83 us at 28x28,
  // batch 256)
  if (nsplit >= 2 && N < 4 * K) return 0;
  *nw_out = nw;
  return nsplit;
}

int pf_conv_stream_groups(int nsplit) { return ST_GRID / nsplit; }

template <int NW, bool PRO, bool BWD, bool MAP, bool AFF = false>
static int stream_launch_t(const ConvArgs& a, int nsplit, hipStream_t st) {
  const int JM = (NW == 256) ? 1 : 2, RS = 16 * JM;
  const size_t aux_fl = PRO ? 2 * (size_t)a.K : (BWD ? 4 * (size_t)NW : 0);
  const size_t lds = (size_t)NW * a.K * 2 + aux_fl * 4 + (size_t)ST_WAVES * RS * (NW + 8) * 2;
  if (int e = pf_require_lds(reinterpret_cast<const void*>(&k_conv1x1_stream<NW, PRO, BWD, MAP, AFF>), lds)) return e;
  k_conv1x1_stream<NW, PRO, BWD, MAP, AFF><<<ST_GRID, ST_THREADS, lds, st>>>(a, nsplit);
  PF_LAUNCH_CHECK();
  return 0;
}

// a: filled by conv_fwd_launch (pf_conv.hip); returns -1 when the stream kernel does not apply
int pf_conv_stream_launch(const ConvArgs& a, bool pro, bool bwd, hipStream_t st) {
  int nw = 0;
  const int nsplit = pf_conv_stream_plan(a.M, a.N, a.K, &nw);
  if (nsplit == 0) return -1;
  const bool map = a.stride != 1;
  if (a.oss != nullptr) {                                  // output affine: stride-1 forward launches (the caller falls back otherwise)
    if (map || bwd) return -1;
#define PF_STA(NWV) do { return pro ? stream_launch_t<NWV, true, false, false, true>(a, nsplit, st)     \
                                    : stream_launch_t<NWV, false, false, false, true>(a, nsplit, st); } while (0)
    if (nw == 64) PF_STA(64);
    if (nw == 128) PF_STA(128);
    PF_STA(256);
#undef PF_STA
  }
#define PF_ST(NWV)                                                                                  \
  do {                                                                                              \
    if (pro) return map ? stream_launch_t<NWV, true, false, true>(a, nsplit, st)                    \
                        : stream_launch_t<NWV, true, false, false>(a, nsplit, st);                  \
    if (bwd) return stream_launch_t<NWV, false, true, false>(a, nsplit, st);                        \
    return map ? stream_launch_t<NWV, false, false, true>(a, nsplit, st)                            \
               : stream_launch_t<NWV, false, false, false>(a, nsplit, st);                          \
  } while (0)
  if (nw == 64) PF_ST(64);
  if (nw == 128) PF_ST(128);
  PF_ST(256);
#undef PF_ST
}
