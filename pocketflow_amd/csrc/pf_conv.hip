// K12 fused with K13/K4: 1x1 convolutions of the student / teacher as MFMA GEMMs over NHWC
// activations, with the batch-norm + ReLU + activation-fake-quant of the PRODUCER layer applied while
// the input tile is staged (prologue) and the residual add + batch-norm statistics of the CONSUMER
// layer computed while the output tile is written (epilogue).
//
//   reference chain per bottleneck block (utils/external/resnet_model.py:257-314 +
//   learners/uniform_quantization/utils.py:51-79), each arrow a separate TF kernel with a full HBM
//   round trip:  x -> BN -> ReLU -> min/max -> quantise -> conv1x1 -> ... -> conv1x1 -> + shortcut
//
//   here:   Y = conv1x1( Q(x) ) [+ R],   Q(x) = fake_quant(act(scale*x + shift))  applied on the fly,
//           partial[g] = per-channel {sum, sumsq, min, max} of the (bf16-rounded) Y tile rows
//   so the activated / quantised tensor Q(x) is never written to HBM and the consumer BN needs no
//   statistics pass:  2 passes over the 256-channel tensors of a block instead of 8.
//
// MI355X mapping: 64-lane wavefronts, v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the A operand
// and the PIXEL tile as the B operand, so that a lane's 4 accumulator values are 4 consecutive output
// channels of one pixel (8-byte LDS writes when the tile is staged for the coalesced 16-byte row
// stores); persistent workgroups (<= 2 per CU) walk row tiles with the next input tile prefetched into
// registers; tiles of one row panel are placed on one XCD (blockIdx % 8) so the panel is read from HBM
// once and shared through that XCD's L2.
//
//   pf_conv1x1_fwd   : forward (student + teacher), also backward-data with the transposed weights
//   pf_conv1x1_wrw   : backward-filter, same prologue on the input operand, split over pixels,
//                      deterministic two-stage reduction (no float atomics)
#include "pf_conv_common.h"

// ---------------------------------------------------------------------------------------------
// forward:  Y[m][n] = sum_k Q(X)[row(m)][k] * W[n][k]  (+ R[m][n]),  optional per-channel stats
// ---------------------------------------------------------------------------------------------
// MAP: strided / remapped rows; the stride-1 instantiations carry no index-division code (it is inlined at every load and
// store site and had pushed the loop body to the size of the instruction cache: 8010 instructions = 64 KiB)
template <int BN_T, bool PRO, bool BWD, bool MAP>
__global__ __launch_bounds__(PF_THREADS, 2) void k_conv1x1_fwd(const ConvArgs a) {
  constexpr int WM = (BN_T == 128) ? 64 : 32;       // pixel rows per wavefront
  constexpr int JM = WM / 16;
  constexpr int CS_LD = BN_T + 8;
  constexpr int VPR = BN_T / 8;                     // 16-byte vectors per output row
  constexpr int RPP = PF_THREADS / VPR;             // rows per write pass
  constexpr int NBV = BN_T * 8 / PF_THREADS;        // weight-tile vectors per thread
  constexpr int A_EL = CV_BM * CV_LDK, B_EL = BN_T * CV_LDK;
  constexpr int C_EL = CV_BM * CS_LD;
  constexpr int RED_FL = 4 * RPP * BN_T;            // stats reduction scratch (floats)
  constexpr int SM_BYTES_AB = (A_EL + B_EL) * 2;
  constexpr int SM_BYTES = SM_BYTES_AB > RED_FL * 4 ? (SM_BYTES_AB > C_EL * 2 ? SM_BYTES_AB : C_EL * 2)
                                                    : RED_FL * 4;
  constexpr int SS_BYTES = PRO ? 2 * CV_MAXK * 4 : 0;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM_BYTES + SS_BYTES];
  float* ssl = reinterpret_cast<float*>(smem + SM_BYTES);     // prologue scale | shift (PRO only)
  __shared__ float bpl[BWD ? 4 * BN_T : 4];                            // bwd-stats mode: scale | shift | mean | invstd of this column tile
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Bs = As + A_EL;
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem);

  // workgroup -> (row-tile group g, column tile tn): the tiles_n column tiles of one row panel sit on
  // one XCD with consecutive dispatch slots
  const int xcd = blockIdx.x & 7, L = blockIdx.x >> 3;
  const int g = xcd + 8 * (L / a.tiles_n), tn = L % a.tiles_n;
  const int n0 = tn * BN_T;
  const int nk = (a.K + CV_BK - 1) / CV_BK;
  const int ntl = (g < a.tiles_m) ? (a.tiles_m - g + a.G - 1) / a.G : 0;
  const int total = ntl * nk;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = (BN_T == 128) ? (wave >> 1) : wave, wn = (BN_T == 128) ? (wave & 1) : 0;
  const int frow = lane & 15, fk = (lane >> 4) * 8;
  const int lrow = tid >> 3, kp = tid & 7;          // staging: 8 lanes cover 128 contiguous bytes of a row

  Pro pro;
  pro_init(pro, a.act_lo, a.act_hi, a.kq, PRO ? a.slot : nullptr);

  f32x4 acc[4][JM];
  // input tiles are prefetched TWO k-steps ahead in two register sets (ra0 / ra1, used alternately): with
  // ~2 workgroups per CU one 16 KiB tile in flight per workgroup covers only half of the bandwidth-latency
  // product of this chip's HBM; the (L2-resident) weight tile stays one step ahead
  uint4 ra0[4], ra1[4], rb[NBV];
  uint32_t av0 = 0, av1 = 0;                        // validity bits of the staged vectors

  auto gloadA = [&](int it, uint4 (&ra)[4], uint32_t& av) {
    const int ti = it / nk, ks = it - ti * nk;
    const int m0 = (g + ti * a.G) * CV_BM;
    const int k = ks * CV_BK + kp * 8;
    av = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + lrow + i * 32;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < a.M && k < a.K) {
        // (caching the mapped rows per tile instead of re-deriving them per load was tried: the extra live registers
        // spill and the strided projections got 25 % slower -- 223 vs 179 us)
        const int64_t r = (!MAP || a.ymap) ? (int64_t)m : map_row(a, m);
        av |= 1u << i;
        v = *reinterpret_cast<const uint4*>(a.X + r * a.K + k);
      }
      ra[i] = v;
    }
  };
  auto gloadB = [&](int it) {
    const int ti = it / nk, ks = it - ti * nk;
    const int k = ks * CV_BK + kp * 8;
#pragma unroll
    for (int i = 0; i < NBV; ++i) {
      const int n = n0 + lrow + i * 32;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n < a.N && k < a.K) v = *reinterpret_cast<const uint4*>(a.W + (int64_t)n * a.K + k);
      rb[i] = v;
    }
  };
  auto sstore = [&](int it, const uint4 (&ra)[4], uint32_t av) {
    if (PRO) {
      const int ti = it / nk, ks = it - ti * nk;
      const int k = ks * CV_BK + kp * 8;
      if (k < a.K) {
        const float4 s0 = *reinterpret_cast<const float4*>(ssl + k), s1 = *reinterpret_cast<const float4*>(ssl + k + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(ssl + a.K + k), h1 = *reinterpret_cast<const float4*>(ssl + a.K + k + 4);
        pro.sc[0] = s0.x; pro.sc[1] = s0.y; pro.sc[2] = s0.z; pro.sc[3] = s0.w;
        pro.sc[4] = s1.x; pro.sc[5] = s1.y; pro.sc[6] = s1.z; pro.sc[7] = s1.w;
        pro.sh[0] = h0.x; pro.sh[1] = h0.y; pro.sh[2] = h0.z; pro.sh[3] = h0.w;
        pro.sh[4] = h1.x; pro.sh[5] = h1.y; pro.sh[6] = h1.z; pro.sh[7] = h1.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint4 v = ra[i];
      if (PRO) v = ((av >> i) & 1u) ? pro_apply(pro, v) : make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(As + (lrow + i * 32) * CV_LDK + kp * 8) = v;
    }
#pragma unroll
    for (int i = 0; i < NBV; ++i)
      *reinterpret_cast<uint4*>(Bs + (lrow + i * 32) * CV_LDK + kp * 8) = rb[i];
  };

  // running per-thread statistics of the columns this thread writes (8 channels x {sum, sumsq, min, max})
  float st_s[8], st_q[8], st_mn[8], st_mx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; st_mn[j] = INFINITY; st_mx[j] = -INFINITY; }
  const int wvec = tid % VPR, wrow = tid / VPR;

  if (PRO) {                                        // scale | shift of all K input channels, once per workgroup
    for (int i = tid; i < 2 * a.K; i += PF_THREADS)     // staged FOLDED (pf_conv_common.h)
      ssl[i] = (i < a.K) ? pro_fold_scale(pro, a.ss[i]) : pro_fold_shift(pro, a.ss[i]);
  }
  constexpr bool bwd_stats = BWD;                    // backward-data + BN-backward statistics (PRO == false)
  if (bwd_stats) {
    for (int i = tid; i < 4 * BN_T; i += PF_THREADS) {
      const int q = i / BN_T, c = n0 + (i - q * BN_T);
      float v = 0.f;
      if (c < a.N) v = (q < 2) ? a.bss[q * a.N + c] : a.bmi[(q - 2) * a.N + c];
      bpl[i] = v;
    }
  }
  if (total > 0) { gloadA(0, ra0, av0); gloadB(0); }
  if (total > 1) gloadA(1, ra1, av1);
  if (PRO || bwd_stats) __syncthreads();

  auto body = [&](int it, uint4 (&ra)[4], uint32_t& av) {
    const int ti = it / nk, ks = it - ti * nk;
    if (ks == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    sstore(it, ra, av);
    __syncthreads();
    if (it + 2 < total) gloadA(it + 2, ra, av);
    if (it + 1 < total) gloadB(it + 1);
#pragma unroll
    for (int kk = 0; kk < CV_BK / 32; ++kk) {
      bf16x8 wf[4], xf[JM];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        wf[i] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 64 + i * 16 + frow) * CV_LDK + kk * 32 + fk);
#pragma unroll
      for (int j = 0; j < JM; ++j)
        xf[j] = *reinterpret_cast<const bf16x8*>(As + (wm * WM + j * 16 + frow) * CV_LDK + kk * 32 + fk);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (ks != nk - 1) return;

    // ---- epilogue of one [128][BN_T] tile ---------------------------------------------------------
    // accumulators: D row = channel (lane >> 4) * 4 + r of block i, D col = pixel (lane & 15) of block j
    const int m0 = (g + ti * a.G) * CV_BM;
    if (!bwd_stats && a.R != nullptr) {
      // residual added to the fp32 accumulators: the sum is rounded to bf16 ONCE (in the staging below).  This lane's
      // four channels of pixel (j, frow) are 8 contiguous bytes of R.
#pragma unroll
      for (int j = 0; j < JM; ++j) {
        const int m = m0 + wm * WM + j * 16 + frow;
        if (m < a.M) {
          const int64_t orow = (MAP && a.ymap) ? map_row(a, m) : (int64_t)m;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
            if (n < a.N) {
              const uint2 r = *reinterpret_cast<const uint2*>(a.R + orow * a.N + n);
              acc[i][j][0] += __uint_as_float(r.x << 16); acc[i][j][1] += __uint_as_float(r.x & 0xFFFF0000u);
              acc[i][j][2] += __uint_as_float(r.y << 16); acc[i][j][3] += __uint_as_float(r.y & 0xFFFF0000u);
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < JM; ++j) {
        const uint2 v = make_uint2(pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3]));
        *reinterpret_cast<uint2*>(Cs + (wm * WM + j * 16 + frow) * CS_LD + wn * 64 + i * 16 + (lane >> 4) * 4) = v;
      }
    // BN-input vectors (backward statistics) of this thread's 8 (or 4) output rows: all loads in flight before the LDS
    // hand-off
    uint4 rres[CV_BM / RPP];
    const bf16_t* __restrict__ side = bwd_stats ? a.bx : nullptr;   // second [M][N] operand of the row pass
    if (side != nullptr) {
#pragma unroll
      for (int p = 0; p < CV_BM / RPP; ++p) {
        const int m = m0 + wrow + p * RPP, n = n0 + wvec * 8;
        rres[p] = make_uint4(0, 0, 0, 0);
        if (m < a.M && n < a.N) {
          const int64_t orow = (MAP && a.ymap) ? map_row(a, m) : (int64_t)m;
          rres[p] = *reinterpret_cast<const uint4*>(side + orow * a.N + n);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < CV_BM / RPP; ++p) {
      const int rl = wrow + p * RPP;
      const int m = m0 + rl, n = n0 + wvec * 8;
      if (m < a.M && n < a.N) {
        uint4 c = *reinterpret_cast<const uint4*>(Cs + rl * CS_LD + wvec * 8);
        const int64_t orow = (MAP && a.ymap) ? map_row(a, m) : (int64_t)m;
        if (bwd_stats) {
          float f[8], xv[8];
          unpack8(c, f);
          unpack8(rres[p], xv);
          const float* bp = bpl + wvec * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float u = fmaf(bp[j], xv[j], bp[BN_T + j]);
            const float dy = (u > a.b_lo && u < a.b_hi) ? f[j] : 0.f;
            st_s[j] += dy;
            st_q[j] = fmaf(dy, (xv[j] - bp[2 * BN_T + j]) * bp[3 * BN_T + j], st_q[j]);
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
        *reinterpret_cast<uint4*>(a.Y + orow * a.N + n) = c;
      }
    }
    __syncthreads();
  };

  for (int it = 0; it < total; it += 2) {
    body(it, ra0, av0);
    if (it + 1 < total) body(it + 1, ra1, av1);
  }

  // ---- per-workgroup statistics -> partial[g][4][N] (fixed order: deterministic) ------------------
  if (a.partial != nullptr) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(0 * RPP + wrow) * BN_T + wvec * 8 + j] = st_s[j];
      red[(1 * RPP + wrow) * BN_T + wvec * 8 + j] = st_q[j];
      red[(2 * RPP + wrow) * BN_T + wvec * 8 + j] = st_mn[j];
      red[(3 * RPP + wrow) * BN_T + wvec * 8 + j] = st_mx[j];
    }
    __syncthreads();
    const int nstat = bwd_stats ? 2 : 4;
    for (int t = tid; t < nstat * BN_T; t += PF_THREADS) {
      const int stat = t / BN_T, c = t - stat * BN_T;
      float v = red[(stat * RPP) * BN_T + c];
      for (int r = 1; r < RPP; ++r) {
        const float w = red[(stat * RPP + r) * BN_T + c];
        v = (stat < 2) ? (v + w) : (stat == 2 ? fminf(v, w) : fmaxf(v, w));
      }
      if (n0 + c < a.N) a.partial[((int64_t)g * nstat + stat) * a.N + n0 + c] = v;
    }
  }
}

#include <stdlib.h>
static int conv_bn_of(int N) {
  if (pf_tuning().conv_bn == 64) return 64;           // PF_CONV_BN=64: tuning override (experiments only)
  return (N % 128 == 0) ? 128 : 64;
}

static int conv_fwd_grid(int tiles_m, int tiles_n, int* G_out) {
  // ~2 workgroups per CU (256 CUs); G is a multiple of 8 so that a row panel's column tiles share an XCD
  int G = 512 / tiles_n;
  G = (G / 8) * 8;
  if (G < 8) G = 8;
  const int need = ((tiles_m + 7) / 8) * 8;
  if (G > need) G = need;
  *G_out = G;
  return G * tiles_n;
}

// pf_conv_stream.hip: the barrier-free variant for kernels that fit the LDS
int pf_conv_stream_plan(int M, int N, int K, int* nw_out);
int pf_conv_stream_groups(int nsplit);
int pf_conv_stream_launch(const ConvArgs& a, bool pro, bool bwd, hipStream_t st);

extern "C" int pf_conv1x1_stats_groups(int M, int N) {
  const int bn = conv_bn_of(N);
  int G;
  conv_fwd_grid((M + CV_BM - 1) / CV_BM, (N + bn - 1) / bn, &G);
  return G;
}

// rows of the [G][stats][N] partial-statistics array pf_conv1x1_fwd / pf_conv1x1_bwd_data_bnstats write for an
// [M][K] x [N][K] problem (depends on which kernel the shape is dispatched to)
// pf_igemm.hip: direct-to-LDS staged GEMM for the prologue-free shapes with a deep contraction
extern "C" int pf_conv2d_stats_groups(int M, int N);
int pf_igemm_stats_groups(int M, int N, int pro);           // the launcher's own tile decision (prologue variant or not)
int pf_igemm_conv1x1(const void* X, const void* W, void* Y, const void* R, float* partial, const void* bn_x,
                     const float* bss, const float* bmi, float b_lo, float b_hi, const float* scale_shift,
                     const uint32_t* slot, float kq, float act_lo, float act_hi, int M, int N, int K, int Ho, int Wo,
                     int H, int Wd, int stride, const float* oss, int oact, hipStream_t st, int ymap = 0);
// Which 1x1 shapes go to the direct-to-LDS staged kernel (after the resident-kernel variant had its pick).  Measured
// (tools/gpu/igemm_bench.py, conv_bench2.py): prologue-free GEMMs win from K = 512 up; with the prologue the in-LDS pass
// behind asynchronous staging beats the register-staged tiles of this file on every shape it was tried on.
#ifndef PF_CONV_IGEMM_PLAIN_MINK
#define PF_CONV_IGEMM_PLAIN_MINK 256     // (variant builds for A/B runs: tools/gpu/build_variant.sh -DPF_CONV_IGEMM_PLAIN_MINK=512 = rounds 2-5)
#endif
static bool conv_use_igemm(bool pro, int K) {
  if ((pro ? pf_tuning().conv_igemm_pro : pf_tuning().conv_igemm) == 0) return false;    // PF_CONV_IGEMM[_PRO]=0: tuning / A-B override
  // (round 6: K >= 256 -- with the teacher's bn3 folded into conv2 its conv3 launches of stage 3, 256 -> 1024 at 14 x 14, became
  // prologue-free and fell to the register-staged tiles of this file at 157 us; the staged kernel runs them in 44)
  return (K % 64) == 0 && (pro || K >= PF_CONV_IGEMM_PLAIN_MINK);
}

extern "C" int pf_conv1x1_stats_groups_k(int M, int N, int K, int prologue) {
  int nw = 0;
  const int nsplit = pf_conv_stream_plan(M, N, K, &nw);
  if (nsplit > 0) return pf_conv_stream_groups(nsplit);
  if (conv_use_igemm(prologue != 0, K)) return pf_igemm_stats_groups(M, N, prologue != 0);
  return pf_conv1x1_stats_groups(M, N);
}

static int conv_fwd_launch(const void* X, const void* W, void* Y, const void* R, const float* scale_shift,
                           int act, const uint32_t* slot, int bits, float* partial, int M, int N, int K,
                           int Ho, int Wo, int H, int Wd, int stride, int ymap, const void* bx,
                           const float* bss, const float* bmi, int bact, void* stream, const float* oss = nullptr,
                           int oact = PF_ACT_NONE) {
  if (oss != nullptr && (R != nullptr || partial != nullptr || bx != nullptr || ymap)) return (int)hipErrorInvalidValue;
  if (M <= 0 || N <= 0 || K <= 0 || (K % 8) || (N % 8)) return (int)hipErrorInvalidValue;
  if (!pf_aligned16(X) || !pf_aligned16(W) || !pf_aligned16(Y) || (R && !pf_aligned16(R)))
    return (int)hipErrorInvalidValue;
  if (slot != nullptr && (bits < 1 || bits > 32 || scale_shift == nullptr)) return (int)hipErrorInvalidValue;
  if (scale_shift != nullptr && K > CV_MAXK) return (int)hipErrorInvalidValue;
  if (stride < 1 || (stride > 1 && (Ho <= 0 || Wo <= 0 || H <= 0 || Wd <= 0))) return (int)hipErrorInvalidValue;
  ConvArgs a;
  a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.Y = (bf16_t*)Y; a.R = (const bf16_t*)R;
  a.ss = scale_shift; a.slot = slot; a.partial = partial;
  a.kq = uq_k_of_bits(slot ? bits : 8);
  a.act_lo = (act == PF_ACT_NONE) ? -INFINITY : 0.0f;
  a.act_hi = (act == PF_ACT_RELU6) ? 6.0f : INFINITY;
  a.M = M; a.N = N; a.K = K;
  a.Ho = Ho; a.Wo = Wo; a.H = H; a.Wd = Wd; a.stride = stride; a.ymap = ymap; a.rows_per_split = 0;
  a.bx = (const bf16_t*)bx; a.bss = bss; a.bmi = bmi;
  a.b_lo = (bact == PF_ACT_NONE) ? -INFINITY : 0.0f;
  a.b_hi = (bact == PF_ACT_RELU6) ? 6.0f : INFINITY;
  a.oss = oss; a.oact = oact;
  if (bx != nullptr && (R != nullptr || partial == nullptr || bss == nullptr || bmi == nullptr || stride != 1 ||
                        !pf_aligned16(bx)))
    return (int)hipErrorInvalidValue;
  const int bn = conv_bn_of(N);
  a.tiles_m = (M + CV_BM - 1) / CV_BM;
  a.tiles_n = (N + bn - 1) / bn;
  const int grid = conv_fwd_grid(a.tiles_m, a.tiles_n, &a.G);
  hipStream_t st = (hipStream_t)stream;
  const bool pro = scale_shift != nullptr;
  if (bx != nullptr && pro) return (int)hipErrorInvalidValue;
  {
    const int r = pf_conv_stream_launch(a, pro, bx != nullptr, st);      // HBM-bound shapes: kernel resident in LDS
    if (r >= 0) return r;
  }
  if (!ymap && conv_use_igemm(pro, K)) {
    const int r = pf_igemm_conv1x1(X, W, Y, R, partial, bx, bss, bmi, a.b_lo, a.b_hi, scale_shift, slot, a.kq, a.act_lo,
                                   a.act_hi, M, N, K, Ho, Wo, H, Wd, stride, oss, oact, st);
    if (r >= 0) return r;
  }
  // backward-data of a strided projection (dense dY rows, output rows scattered to (ho * stride, wo * stride) of a pre-zeroed dX):
  // the staged kernel with its row scatter (round 6; the register-staged tiles of this file ran these at 106 us)
  if (ymap && stride > 1 && !pro && bx == nullptr && R == nullptr && partial == nullptr && oss == nullptr && conv_use_igemm(false, K) &&
      (N % 64) == 0) {
    const int r = pf_igemm_conv1x1(X, W, Y, nullptr, nullptr, nullptr, nullptr, nullptr, a.b_lo, a.b_hi, nullptr, nullptr, a.kq, a.act_lo,
                                   a.act_hi, M, N, K, Ho, Wo, H, Wd, stride, nullptr, PF_ACT_NONE, st, 1);
    if (r >= 0) return r;
  }
  if (oss != nullptr) {
    // no kernel with the folded pass took the shape: the plain launch, then the stand-alone pass IN PLACE (same result, one more
    // read + write of Y)
    const int r = conv_fwd_launch(X, W, Y, R, scale_shift, act, slot, bits, partial, M, N, K, Ho, Wo, H, Wd, stride, ymap, bx, bss,
                                  bmi, bact, stream);
    if (r != 0) return r;
    return pf_bn_act_quant_apply(Y, Y, PF_BF16, M, N, oss, oact, nullptr, 8, 0, stream);
  }
  const bool map = stride != 1;
#define PF_CV(BNV)                                                                             \
  do {                                                                                         \
    if (pro) { if (map) k_conv1x1_fwd<BNV, true, false, true><<<grid, PF_THREADS, 0, st>>>(a);  \
               else k_conv1x1_fwd<BNV, true, false, false><<<grid, PF_THREADS, 0, st>>>(a); }   \
    else if (bx != nullptr) k_conv1x1_fwd<BNV, false, true, false><<<grid, PF_THREADS, 0, st>>>(a); \
    else { if (map) k_conv1x1_fwd<BNV, false, false, true><<<grid, PF_THREADS, 0, st>>>(a);     \
           else k_conv1x1_fwd<BNV, false, false, false><<<grid, PF_THREADS, 0, st>>>(a); }      \
  } while (0)
  if (bn == 128) PF_CV(128); else PF_CV(64);
#undef PF_CV
  PF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pf_conv1x1_fwd(const void* X, const void* W, void* Y, const void* R, const float* scale_shift,
                              int act, const uint32_t* slot, int bits, float* partial, int M, int N, int K,
                              int Ho, int Wo, int H, int Wd, int stride, int ymap, void* stream) {
  return conv_fwd_launch(X, W, Y, R, scale_shift, act, slot, bits, partial, M, N, K, Ho, Wo, H, Wd, stride, ymap,
                         nullptr, nullptr, nullptr, PF_ACT_NONE, stream);
}

// pf_conv1x1_fwd with the CONSUMER's inference-mode BN + activation folded into the row pass of the epilogue (round 6):
// Y = act_out(out_scale[n] * bf16(conv) + out_shift[n]) -- what pf_bn_act_quant_apply (quantize = 0) would make of the stored
// output, bit for bit, without the write + read of the raw output in between.  No residual, no statistics (an inference-mode
// BN needs none).  The teacher of the distillation step (learners/distillation_helper.py:60-84: forward_eval) runs through it.
extern "C" int pf_conv1x1_fwd_affine(const void* X, const void* W, void* Y, const float* scale_shift, int act,
                                     const float* out_scale_shift, int out_act, int M, int N, int K, int Ho, int Wo, int H,
                                     int Wd, int stride, void* stream) {
  if (out_scale_shift == nullptr) return (int)hipErrorInvalidValue;
  return conv_fwd_launch(X, W, Y, nullptr, scale_shift, act, nullptr, 8, nullptr, M, N, K, Ho, Wo, H, Wd, stride, 0,
                         nullptr, nullptr, nullptr, PF_ACT_NONE, stream, out_scale_shift, out_act);
}

// backward-data of a stride-1 1x1 convolution, dQ[M][K] = dY[M][N] * W[N][K] (Wt = the transposed kernel
// [K][N]), with the BN-backward statistics of the layer that produced Q in the epilogue:
// partial[G][2][K] = {sum dy, sum dy*xhat} over the rows of each workgroup, G = pf_conv1x1_stats_groups(M, K).
extern "C" int pf_conv1x1_bwd_data_bnstats(const void* dY, const void* Wt, void* dQ, const void* bn_x,
                                           const float* bn_scale_shift, const float* bn_mean_invstd, int bn_act,
                                           float* partial, int M, int N, int K, void* stream) {
  return conv_fwd_launch(dY, Wt, dQ, nullptr, nullptr, PF_ACT_NONE, nullptr, 8, partial, M, K, N, 0, 0, 0, 0, 1, 0,
                         bn_x, bn_scale_shift, bn_mean_invstd, bn_act, stream);
}

// ---------------------------------------------------------------------------------------------
// backward-filter:  dW[n][k] = sum_m dY[m][n] * Q(X)[row(m)][k]
// Output tile 128 (n) x 64 (k) per workgroup and pixel split; both operands are contracted over the
// pixel index, which is the SLOW index of both NHWC tensors, so the tiles are transposed while they
// are staged: LDS image [channel][pixel] with the 16-byte pixel groups XOR-swizzled by the channel
// group (2-way instead of 16-way bank conflicts on the scattered 2-byte writes; fragment reads stay
// single aligned 16-byte reads).
// ---------------------------------------------------------------------------------------------
#define WR_TN 128
#define WR_TK 64
#define WR_BM 64
#define WR_LDM (WR_BM + 8)

__device__ __forceinline__ void st_transposed(bf16_t* S, int c0, int m, const uint4& v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  const int mg = m >> 3, ml = m & 7;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    S[c * WR_LDM + (((mg ^ ((c >> 3) & 7)) << 3) | ml)] = (bf16_t)((w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
  }
}

template <bool PRO>
__global__ __launch_bounds__(PF_THREADS, 2) void k_conv1x1_wrw(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t Ds[WR_TN * WR_LDM];   // dY tile, [n][m]
  __shared__ __attribute__((aligned(16))) bf16_t Qs[WR_TK * WR_LDM];   // Q tile,  [k][m]
  // XCD-aware order: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs; give every
  // XCD a contiguous run of logical ids so that all (n, k) output tiles of one pixel split -- which read the
  // same dY / X rows -- share that XCD's L2 instead of re-reading the rows from HBM on 8 different L2s
  const int ntile = gridDim.x, nwg = gridDim.x * gridDim.y;
  int wg = blockIdx.y * gridDim.x + blockIdx.x;
  {
    const int xcd = wg & 7, idx = wg >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile = wg % ntile, split = wg / ntile;
  const int tn = tile % a.tiles_n, tk = tile / a.tiles_n;
  const int n0 = tn * WR_TN, k0 = tk * WR_TK;
  const int mbeg = split * a.rows_per_split;
  const int mend = (mbeg + a.rows_per_split < a.M) ? (mbeg + a.rows_per_split) : a.M;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int frow = lane & 15, fm = lane >> 4;               // fragment: channel row, 8-pixel group

  Pro pro;
  pro_init(pro, a.act_lo, a.act_hi, a.kq, PRO ? a.slot : nullptr);
  // staging maps: dY tile 64 x 128 = 1024 vectors (4 / thread): pixel = p >> 4, channel group = p & 15
  //               X  tile 64 x 64  =  512 vectors (2 / thread): pixel = p >> 3, channel group = p & 7
  const int qkp = tid & 7;
  if (PRO) {
    const int k = k0 + qkp * 8;
    if (k < a.K) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { pro.sc[j] = pro_fold_scale(pro, a.ss[k + j]); pro.sh[j] = pro_fold_shift(pro, a.ss[a.K + k + j]); }
    }
  }
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 rd[4], rq[2];
  bool vq[2];
  auto gload = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = tid + i * PF_THREADS;
      const int m = mb + (p >> 4), n = n0 + (p & 15) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < mend && n < a.N) v = *reinterpret_cast<const uint4*>(a.W + (int64_t)m * a.N + n);
      rd[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = tid + i * PF_THREADS;
      const int m = mb + (p >> 3), k = k0 + (p & 7) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      vq[i] = (m < mend && k < a.K);
      if (vq[i]) v = *reinterpret_cast<const uint4*>(a.X + map_row(a, m) * a.K + k);
      rq[i] = v;
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = tid + i * PF_THREADS;
      st_transposed(Ds, (p & 15) * 8, p >> 4, rd[i]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = tid + i * PF_THREADS;
      uint4 v = rq[i];
      if (PRO) v = vq[i] ? pro_apply(pro, v) : make_uint4(0, 0, 0, 0);
      st_transposed(Qs, (p & 7) * 8, p >> 3, v);
    }
  };

  if (mbeg < mend) gload(mbeg);
  for (int mb = mbeg; mb < mend; mb += WR_BM) {
    sstore();
    __syncthreads();
    if (mb + WR_BM < mend) gload(mb + WR_BM);
#pragma unroll
    for (int kk = 0; kk < WR_BM / 32; ++kk) {
      bf16x8 df[2], qf[4];
      const int mg = kk * 4 + fm;                           // 8-pixel group of this lane's fragment
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = wave * 32 + i * 16 + frow;
        df[i] = *reinterpret_cast<const bf16x8*>(Ds + c * WR_LDM + ((mg ^ ((c >> 3) & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = j * 16 + frow;
        qf[j] = *reinterpret_cast<const bf16x8*>(Qs + c * WR_LDM + ((mg ^ ((c >> 3) & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[i], qf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D row = n (lane >> 4) * 4 + r of block i, D col = k (lane & 15) of block j
  float* out = a.partial + (int64_t)split * a.N * a.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wave * 32 + i * 16 + (lane >> 4) * 4 + r;
        if (n < a.N && k < a.K) out[(int64_t)n * a.K + k] = acc[i][j][r];
      }
    }
}

// dW = sum over splits, written as float32 or bf16.  One workgroup owns 64 consecutive outputs and one
// of gridDim.y split ranges; its 4 wavefronts take interleaved splits (256-byte coalesced rows) and are
// combined through LDS in a fixed order; a second launch folds the gridDim.y range sums.  Deterministic.
template <typename TO>
__global__ __launch_bounds__(PF_THREADS) void k_wrw_reduce(const float* __restrict__ partial, int splits,
                                                           int64_t n, TO* __restrict__ out, int64_t out_stride) {
  __shared__ float l[4][64];
  const int e_l = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + e_l;
  const int per = (splits + gridDim.y - 1) / gridDim.y;
  const int s0 = blockIdx.y * per, s1 = (s0 + per < splits) ? (s0 + per) : splits;
  float acc = 0.f;
  if (e < n)
    for (int s = s0 + sg; s < s1; s += 4) acc += partial[(int64_t)s * n + e];
  l[sg][e_l] = acc;
  __syncthreads();
  if (sg == 0 && e < n) store_one<TO>(out + (int64_t)blockIdx.y * out_stride + e, (l[0][e_l] + l[1][e_l]) + (l[2][e_l] + l[3][e_l]));
}

// pf_wrw.hip: backward-filter on transposed LDS reads (ds_read_b64_tr_b16), barrier-free main loop
int pf_wrw_tr_splits_1x1(int M, int N, int C);
int pf_wrw2_splits(int M, int N, int C, int taps);
int pf_wrw2_launch(const void* dY, const void* X, float* slabs, const float* scale_shift, int act, const uint32_t* slot,
                   int bits, int M, int N, int C, int th, int tw, int H, int Wd, int Ho, int Wo, int stride, int pad_h,
                   int pad_w, int S, int64_t x_rows, hipStream_t st);
int pf_wrw_tr_launch(const void* dY, const void* X, float* slabs, const float* scale_shift, int act,
                     const uint32_t* slot, int bits, int M, int N, int C, int th, int tw, int H, int Wd, int Ho, int Wo,
                     int stride, int pad_h, int pad_w, int S, hipStream_t st);

// sum of the S split slabs [S][n] -> dW (float32 / bf16), fixed order; workspace holds (S + 32) * n floats
int pf_wrw_reduce(float* workspace, int S, int64_t n, void* dW, int dw_dtype, hipStream_t st) {
  const int gx = (int)((n + 63) / 64);
  int ry = 1;                                           // split ranges of the first reduction stage
  while (ry < 32 && gx * ry < 1024 && ry * 8 <= S) ry *= 2;
  float* stage = workspace + (int64_t)S * n;            // [ry][n] floats behind the split slabs
  if (ry > 1) {
    k_wrw_reduce<float><<<dim3(gx, ry), PF_THREADS, 0, st>>>(workspace, S, n, stage, n);
    if (dw_dtype == PF_F32) k_wrw_reduce<float><<<dim3(gx, 1), PF_THREADS, 0, st>>>(stage, ry, n, (float*)dW, 0);
    else if (dw_dtype == PF_BF16) k_wrw_reduce<bf16_t><<<dim3(gx, 1), PF_THREADS, 0, st>>>(stage, ry, n, (bf16_t*)dW, 0);
    else return (int)hipErrorInvalidValue;
  } else {
    if (dw_dtype == PF_F32) k_wrw_reduce<float><<<dim3(gx, 1), PF_THREADS, 0, st>>>(workspace, S, n, (float*)dW, 0);
    else if (dw_dtype == PF_BF16) k_wrw_reduce<bf16_t><<<dim3(gx, 1), PF_THREADS, 0, st>>>(workspace, S, n, (bf16_t*)dW, 0);
    else return (int)hipErrorInvalidValue;
  }
  PF_LAUNCH_CHECK();
  return 0;
}

// number of pixel splits; the workspace must hold (splits + 32) * N * K floats
extern "C" int pf_conv1x1_wrw_splits(int M, int N, int K) {
  {
    const int s2 = pf_wrw_tr_splits_1x1(M, N, K);
    if (s2 > 0) return s2;
  }
  const int tiles = ((N + WR_TN - 1) / WR_TN) * ((K + WR_TK - 1) / WR_TK);
  int S = (768 + tiles - 1) / tiles;
  const int maxS = (M + 4 * WR_BM - 1) / (4 * WR_BM);       // at least 4 pixel steps per workgroup
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  int rows = (M + S - 1) / S;
  rows = ((rows + WR_BM - 1) / WR_BM) * WR_BM;
  return (M + rows - 1) / rows;
}

extern "C" int pf_conv1x1_wrw(const void* dY, const void* X, void* dW, int dw_dtype, float* workspace,
                              const float* scale_shift, int act, const uint32_t* slot, int bits, int M, int N,
                              int K, int Ho, int Wo, int H, int Wd, int stride, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % 8) || (N % 8) || ((int64_t)N * K) % 4) return (int)hipErrorInvalidValue;
  if (!pf_aligned16(dY) || !pf_aligned16(X) || !pf_aligned16(workspace) || !pf_aligned16(dW))
    return (int)hipErrorInvalidValue;
  if (slot != nullptr && (bits < 1 || bits > 32 || scale_shift == nullptr)) return (int)hipErrorInvalidValue;
  ConvArgs a;
  a.X = (const bf16_t*)X; a.W = (const bf16_t*)dY; a.Y = nullptr; a.R = nullptr;
  a.ss = scale_shift; a.slot = slot; a.partial = workspace;
  a.kq = uq_k_of_bits(slot ? bits : 8);
  a.act_lo = (act == PF_ACT_NONE) ? -INFINITY : 0.0f;
  a.act_hi = (act == PF_ACT_RELU6) ? 6.0f : INFINITY;
  a.M = M; a.N = N; a.K = K;
  a.Ho = Ho; a.Wo = Wo; a.H = H; a.Wd = Wd; a.stride = stride < 1 ? 1 : stride; a.ymap = 0;
  a.bx = nullptr; a.bss = nullptr; a.bmi = nullptr; a.b_lo = 0.f; a.b_hi = INFINITY;
  a.tiles_m = 0; a.G = 0;
  a.tiles_n = (N + WR_TN - 1) / WR_TN;
  const int tiles_k = (K + WR_TK - 1) / WR_TK;
  const int S = pf_conv1x1_wrw_splits(M, N, K);
  hipStream_t st = (hipStream_t)stream;
  const int s_tr = pf_wrw_tr_splits_1x1(M, N, K);
  int done = -1;
  if (s_tr > 0) {
    if (pf_wrw2_splits(M, N, K, 1) > 0) {
      const int64_t x_rows = (a.stride > 1) ? (int64_t)(M / (Ho * Wo)) * H * Wd : (int64_t)M;
      done = pf_wrw2_launch(dY, X, workspace, scale_shift, act, slot, bits, M, N, K, 1, 1, H, Wd, Ho, Wo, a.stride, 0, 0,
                            s_tr, x_rows, st);
    }
    if (done < 0)
      done = pf_wrw_tr_launch(dY, X, workspace, scale_shift, act, slot, bits, M, N, K, 1, 1, H, Wd, Ho, Wo, a.stride, 0, 0,
                              s_tr, st);
  }
  if (done > 0) return done;
  if (done < 0) {
    int rows = (M + S - 1) / S;
    rows = ((rows + WR_BM - 1) / WR_BM) * WR_BM;
    a.rows_per_split = rows;
    dim3 grid(a.tiles_n * tiles_k, S);
    if (scale_shift != nullptr) k_conv1x1_wrw<true><<<grid, PF_THREADS, 0, st>>>(a);
    else k_conv1x1_wrw<false><<<grid, PF_THREADS, 0, st>>>(a);
    PF_LAUNCH_CHECK();
  }
  return pf_wrw_reduce(workspace, (done == 0) ? s_tr : S, (int64_t)N * K, dW, dw_dtype, st);
}
