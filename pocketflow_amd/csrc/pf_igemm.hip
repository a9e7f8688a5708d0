// K12: dense convolutions of the student / teacher as implicit GEMMs on the matrix cores -- forward and
// backward-data of the RxS convolutions (3x3 of the ResNet blocks, utils/external/resnet_model.py:92-103,
// conv2d_fixed_padding) and the plain 1x1 GEMMs that carry no prologue (backward-data of the 1x1 convolutions of
// stages 3-4).  bf16 NHWC activations, KRSC kernels, fp32 accumulation.
//
//   Y[m][n] = sum_{tap, c} X[pix(m, tap)][c] * W[n][tap][c]          m = (img, ho, wo),  tap = (r, s)
//   pix(m, tap) = (img, ho*stride + r - pad_h, wo*stride + s - pad_w); taps that fall outside the image contribute 0
//   (the reference pads the ACTIVATED / quantised tensor with zeros, so the zeros are exact).
//   epilogue, as in pf_conv.hip: + residual, per-channel {sum, sumsq, min, max} of the stored tile (the statistics the
//   consumer BN needs), or -- backward-data -- the BN-backward sums {sum dy, sum dy*xhat} of the producer BN.
//
// MI355X mapping.  The contraction index runs over (tap, 64-channel step); for one step the input tile is BM pixel
// rows x 128 contiguous bytes and the kernel tile BN rows x 128 bytes.  Both go from global memory STRAIGHT INTO LDS
// (global_load_lds, 16 bytes per lane, no staging registers, no ds_write pass): the LDS destination of that
// instruction is lane-linear, so a lane fetches the 16-byte group that belongs at its LDS position -- the gather of
// the implicit im2col (per-lane source rows, a zero row for padding taps) and the XOR swizzle that makes the
// fragment reads bank-conflict free are both applied to the SOURCE address.  Two LDS stages: the loads of step t+1
// are in flight while step t is multiplied, one workgroup barrier per step.  Wavefronts tile the block 2-D
// (64 x 64 accumulators each: 16 fragment reads per 32 MFMAs); weights are MFMA operand A and pixels operand B so
// that a lane's four accumulator values are consecutive output channels of one pixel (8-byte writes when the tile is
// transposed through LDS for 16-byte row stores).  Persistent workgroups walk row tiles, the column tiles of one row
// panel share an XCD (blockIdx % 8) and therefore that XCD's L2.
#include "pf_conv_common.h"
#include <stdio.h>
#include <stdlib.h>

#include "pf_igemm.h"

// Ablation builds for tools/gpu/igemm_ablate.py ONLY (never in libpocketflow_hip.so): -DPF_IG_ABLATE=1 drops the MFMAs and
// their fragment reads (what is left is the LDS-DMA fill + barriers), =2 drops the LDS-DMA (matrix work + fragment reads +
// barriers), =3 drops both the LDS-DMA and the fragment reads (matrix work + barriers); =4 is 3 WITHOUT THE EPILOGUE (no C
// staging, no statistics, no stores: the main loop's matrix work and barriers alone), =5 the complete main loop without the
// epilogue (full - 5 = what the epilogue costs).  Results are garbage by design.
#ifndef PF_IG_ABLATE
#define PF_IG_ABLATE 0
#endif
#define PF_IG_NO_DMA (PF_IG_ABLATE == 2 || PF_IG_ABLATE == 3 || PF_IG_ABLATE == 4)
#define PF_IG_FAKE_FRAGS (PF_IG_ABLATE == 3 || PF_IG_ABLATE == 4)

// -DPF_IG_TIMING (tools/gpu/igemm_timeline.py only, never in libpocketflow_hip.so): wavefront 0 of the middle workgroup keeps
// s_memtime stamps of its THIRD tile in scalar registers and writes them to `a.zero` when the tile is done (stores inside the tile
// would ride on vmcnt and falsify the counted waits).
#ifdef PF_IG_TIMING
#define PF_IG_STAMP(k) do { if (tm_tile == 2) tmk[k] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
#define PF_IG_STAMP(k) do { } while (0)
#endif

// (A wave-specialised MODE 3 -- producer wavefronts for the LDS-DMA and the prologue pass, consumer wavefronts for the MFMAs -- lived
// here in rounds 3-4: measured no faster than the single-role three-stage kernel, profiles/r03_pro_bench.txt, and removed.)
// SUB: the launch walks a sub-grid of a larger kernel buffer and scatters its rows (IgArgs.w_* / o_*: strided backward-data by parity
// classes).  A template parameter, not a run-time branch: the extra scalar state cost the ordinary kernels 16-24 more spilled SGPRs
// (v_writelane / v_readlane traffic in the staging code) when it was one.
// AFF: the output affine of the row pass (IgArgs.oss: the consumer's inference-mode BN + activation) -- compiled in for the forward
// launches of an inference-mode network only
// Experiment of round 6 (tools/gpu/a8_codes_probe.py, profiles/r06_a8_codes_ab.txt), compiled into a VARIANT library only
// (tools/gpu/build_variant.sh codes -DPF_IG_CODES): the input operand of the forward kernel is one BYTE per element -- the code
// c in 0..255 of an 8-bit activation grid x = alpha * c -- staged by the same LDS-DMA at half the bytes (64-byte rows, four
// 16-byte groups, group position XOR (row >> 2) & 3: the 8-byte fragment reads of a half wave touch 64 distinct banks),
// converted to bf16 in the fragment path (v_cvt_f32_ubyte + v_perm: integers below 256 are exact in bf16) and scaled by alpha
// on the accumulators.  The product library never defines PF_IG_CODES.
#ifdef PF_IG_CODES
#define PF_IG_XB 1
__device__ __forceinline__ bf16x8 ig_codes_to_bf16x8(uint2 c) {
  uint32_t r[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t w = h ? c.y : c.x;
    const float f0 = (float)(w & 0xffu), f1 = (float)((w >> 8) & 0xffu), f2 = (float)((w >> 16) & 0xffu), f3 = (float)(w >> 24);
    r[h * 2 + 0] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);   // upper halves: exact bf16
    r[h * 2 + 1] = __builtin_amdgcn_perm(__float_as_uint(f3), __float_as_uint(f2), 0x07060302u);
  }
  const uint4 u = make_uint4(r[0], r[1], r[2], r[3]);
  return *reinterpret_cast<const bf16x8*>(&u);
}
#else
#define PF_IG_XB 2
#endif

template <int BM, int BN, int WM, int WN, int NS, int MODE, bool SUB = false, bool AFF = false>
__global__ __launch_bounds__(64 * WM * WN) void k_igemm(const IgArgs a) {
  constexpr bool BWD = (MODE == IG_BWD), PRO = (MODE == IG_PRO);
  static_assert(!PRO || NS == 2 || NS == 3, "the in-LDS prologue pass is written for the 2- and 3-stage rings");
  // PRO with three stages: the loads of step ks+2 travel, the lanes transform their own vectors of step ks+1 in LDS and the
  // matrix cores multiply step ks -- all inside ONE barrier interval (round 2's two-stage form ran them back to back:
  // wait for the loads, transform, barrier, multiply; measured 5.5 k cycles per step against 0.5 k of MFMA issue).
  constexpr bool PRO3 = PRO && NS == 3;
  constexpr int NWC = WM * WN;                      // wavefronts
  constexpr int T = 64 * NWC;                       // threads of the workgroup
  constexpr int TS = T;                             // threads that STAGE
  constexpr int TE = T;                             // threads of the epilogue row stores
  constexpr int WR = BM / WM, WC = BN / WN;         // wavefront tile: pixels x channels
  constexpr int JM = WR / 16, NI = WC / 16;
  constexpr int XB = PF_IG_XB;                      // bytes per input element (1: the codes experiment)
  constexpr int ACH = 4 * XB, AROW = 64 * XB;       // 16-byte groups / bytes of an input-tile row (64 channels)
  constexpr int AS = BM * ACH / TS, BS = BN * 8 / TS; // 16-byte loads per staging lane and step (input / kernel tile)
  constexpr int A_BYTES = BM * AROW, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int CS_LD = BN + 8, CS_LD_B = BN + 8;
  constexpr int VPR = BN / 8, RPP = TE / VPR, NP = BM / RPP;
  static_assert(AS >= 1 && BS >= 1 && NP >= 1, "tile too small for the block");
  constexpr int LPS = AS + BS;                      // LDS-DMA instructions per lane and stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NS stages; aliased: C tile, statistics scratch
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem);
  constexpr int RING_B = NS * STAGE, CT_B = BM * CS_LD_B * 2;
  float* bpl = reinterpret_cast<float*>(smem + (RING_B > CT_B ? RING_B : CT_B));   // BWD: scale | shift | mean | invstd [4][BN]
  float* ssl = bpl;                                                                // PRO3: FOLDED scale | shift [2][C] of the producer BN

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);              // scalar: LDS-DMA bases stay in SGPRs
  const int swave = wave;                                                 // index among the staging wavefronts
  const int cwave = wave % NWC;
  const int wm = cwave / WN, wn = cwave % WN;
  const int l15 = lane & 15, q = lane >> 4;
  const int srow = (swave * 64 + lane) >> 3;                              // staging row of this lane (per 16-byte slot)
  const int schunk = (lane & 7) ^ ((lane >> 3) & 7);                      // source 16-byte group for its LDS position
  const int srowA = (XB == 2) ? srow : ((swave * 64 + lane) >> 2);        // the same for the input tile (codes: 4 groups per row)
  const int schunkA = (XB == 2) ? schunk : ((lane & 3) ^ ((lane >> 4) & 3));

  const int xcd = blockIdx.x & 7, L = blockIdx.x >> 3;
  const int g = xcd + 8 * (L / a.tiles_n), tn = L % a.tiles_n;
  const int n0 = tn * BN;
  const int cch = a.C >> 6;                                               // 64-channel steps per tap
  const int taps = a.th * a.tw;
  const int nk = taps * cch;
  const int64_t wrow = (int64_t)(SUB ? a.w_taps_full : taps) * a.C;       // kernel row length (elements) of the buffer that is walked
  const int hw_o = a.Ho * a.Wo;

  if (BWD) {
    for (int i = tid; i < 4 * BN; i += T) {
      const int qq = i / BN, c = n0 + (i - qq * BN);
      float v = 0.f;
      if (c < a.N) v = (qq < 2) ? a.bss[qq * a.N + c] : a.bmi[(qq - 2) * a.N + c];
      bpl[i] = v;
    }
  }

  float st_s[8], st_q[8], st_mn[8], st_mx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; st_mn[j] = INFINITY; st_mx[j] = -INFINITY; }
  const int wvec = tid % VPR, wrw = tid / VPR;
  const bf16_t* __restrict__ side = BWD ? a.bx : nullptr;                 // the residual is added on the accumulators
  Pro pro;
  pro_init(pro, a.act_lo, a.act_hi, a.kq, PRO ? a.slot : nullptr);
  if (PRO3) {
    // ordinary loads, BEFORE any LDS-DMA is in flight: beside a pending buffer_load ... lds the compiler waits vmcnt(0) for
    // every register-destination load, which would drain the ring once per step
    for (int i = tid; i < 2 * a.C; i += T) ssl[i] = (i < a.C) ? pro_fold_scale(pro, a.ss[i]) : pro_fold_shift(pro, a.ss[i]);
    __syncthreads();
  }

  // Buffer descriptors (wave-uniform): LDS-DMA through buffer_load ... lds takes a 32-bit per-lane byte offset plus a
  // scalar offset, and a lane whose offset lies outside the buffer gets ZEROS -- which is exactly what padding taps,
  // tail rows and tail channels need.  No zero page, no 64-bit per-lane address arithmetic in the main loop.
  const pf_rsrc_t rsX = PF_MAKE_RSRC(a.X, a.x_bytes);
  const pf_rsrc_t rsW = PF_MAKE_RSRC(a.W, a.w_bytes);
  constexpr uint32_t OOB = 0x80000000u;
  // kernel-tile source rows of this lane (fixed for the whole launch)
  uint32_t boff[BS];
#pragma unroll
  for (int i = 0; i < BS; ++i) {
    const int n = n0 + i * (TS / 8) + srow;
    boff[i] = (n < a.N) ? (uint32_t)(((int64_t)n * wrow + schunk * 8) * 2) : OOB;
  }

  const bool pointwise = a.th == 1 && a.tw == 1 && a.stride == 1 && a.pad_h == 0 && a.pad_w == 0 && a.Ho == a.H && a.Wo == a.Wd;
  // Cross-tile prefetch (three-stage kernels, forward / plain modes): the C staging of the epilogue aliases ring buffers 0 and 1
  // only, so every tile starts its ring at buffer RB0 = 2 and the FIRST stage of the next tile is issued into that buffer right
  // behind the last k-step's barrier -- its latency (~2 us, a seventh of a 4-step tile) travels under the epilogue.  The epilogue
  // of such a kernel touches LDS through the asm helpers only (an ordinary LDS access beside a pending LDS-DMA makes hipcc wait
  // vmcnt(0)) and uses raw barriers.
  constexpr bool XPRE = !BWD && NS == 3 && 2 * STAGE >= BM * CS_LD_B * 2;
  constexpr int RB0 = XPRE ? 2 : 0;
  // input rows of this lane: byte offset of the top-left input pixel of the receptive field (may lie outside the
  // image: the sum with the tap offset is only used for taps whose bit is set) and the mask of taps inside the image
  uint32_t pbase[AS], pmask[AS];
  // the steps are staged in order: running (tap, tap row, tap column, channel step) instead of divisions
  int s_r = 0, s_s = 0, s_cc = 0, s_ks = 0, s_tap = 0;
  auto setup_tile = [&](int m0) {
#pragma unroll
    for (int i = 0; i < AS; ++i) {
      const int m = m0 + i * (TS / ACH) + srowA;
      pbase[i] = 0; pmask[i] = 0;
      if (pointwise) {
        // 1x1, stride 1, no padding (every 1x1 layer of the step): input row == output row, one tap -- no divisions, no tap
        // loop (the generic branch costs ~300 instructions per tile, 5 % of a 4-step tile)
        if (m < a.M) { pbase[i] = (uint32_t)(m * a.C + schunkA * (16 / XB)) * (uint32_t)XB; pmask[i] = 1u; }
      } else if (m < a.M) {
        const int img = m / hw_o, rem = m - img * hw_o;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        const int h0 = ho * a.stride - a.pad_h, w0 = wo * a.stride - a.pad_w;
        pbase[i] = (uint32_t)(((img * a.H + h0) * a.Wd + w0) * a.C + schunkA * (16 / XB)) * (uint32_t)XB;   // modulo 2^32 on purpose
        uint32_t mk = 0;
        for (int r = 0; r < a.th; ++r)
          for (int sx = 0; sx < a.tw; ++sx)
            if ((unsigned)(h0 + r) < (unsigned)a.H && (unsigned)(w0 + sx) < (unsigned)a.Wd) mk |= 1u << (r * a.tw + sx);
        pmask[i] = mk;
      }
    }
    s_r = 0; s_s = 0; s_cc = 0; s_ks = 0; s_tap = 0;
  };
  auto stage = [&](int buf) {
    unsigned char* As = smem + buf * STAGE;
    unsigned char* Bs = As + A_BYTES;
    const uint32_t tapoff = (uint32_t)(((s_r * a.Wd + s_s) * a.C + s_cc * 64) * XB);   // wave-uniform
#if !PF_IG_NO_DMA
#pragma unroll
    for (int i = 0; i < AS; ++i) {
      const uint32_t voff = ((pmask[i] >> s_tap) & 1u) ? (pbase[i] + tapoff) : OOB;
      PF_BUFFER_LOAD_LDS16(rsX, As + (i * (TS / ACH) + swave * (64 / ACH)) * AROW, voff, 0);
    }
    // byte offset of this step's (tap, 64-channel group) inside a kernel row: s_ks * 128 when the launch walks its own kernel
    uint32_t woff = (uint32_t)(s_ks * 128);                                             // wave-uniform
    if constexpr (SUB) woff = (uint32_t)((((a.w_r0 + s_r * a.w_rs) * a.w_S + a.w_s0 + s_s * a.w_ss) * a.C + s_cc * 64) * 2);
#pragma unroll
    for (int i = 0; i < BS; ++i)
      PF_BUFFER_LOAD_LDS16(rsW, Bs + (i * (TS / 8) + swave * 8) * 128, boff[i], woff);
#else
    (void)As; (void)Bs; (void)tapoff;
#endif
    ++s_ks;
    if (++s_cc == cch) { s_cc = 0; ++s_tap; if (++s_s == a.tw) { s_s = 0; ++s_r; } }
  };
  bool pre = false;                                                         // XPRE: stage 0 of this tile is already in flight
#ifdef PF_IG_TIMING
  uint32_t tmk[16];
  int tm_tile = -1;
#pragma unroll
  for (int k = 0; k < 16; ++k) tmk[k] = 0;
#endif
  for (int tm = g; tm < a.tiles_m; tm += a.G) {
    const int m0 = tm * BM;
#ifdef PF_IG_TIMING
    ++tm_tile;
#endif
    PF_IG_STAMP(0);
    if (!pre) setup_tile(m0);

    f32x4 acc[NI][JM];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < JM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // residual add on the fp32 accumulators (ONE rounding to bf16, in the staging below): this lane's four columns of
    // row (j, l15) are 8 contiguous bytes of R -- 32-byte segments per row, which only the pre-activation networks
    // (shortcut added to a raw convolution) pay.  The loads are issued at the top of the LAST k-step and travel under its
    // MFMAs (issued in the epilogue they cost 10-13 us per launch on the 14x14 / 7x7 conv3 layers).
    uint2 rr[NI][JM];
    const bool has_r = !BWD && a.R != nullptr;
    constexpr bool RPRE = BM * BN <= 128 * 256;                       // larger tiles have no 32 spare registers: load at use
    auto load_residual = [&]() {
#pragma unroll
      for (int j = 0; j < JM; ++j) {
        const int m = m0 + wm * WR + j * 16 + l15;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int n = n0 + wn * WC + i * 16 + q * 4;
          rr[i][j] = make_uint2(0, 0);
          if (m < a.M && n < a.N) rr[i][j] = *reinterpret_cast<const uint2*>(a.R + (int64_t)m * a.N + n);
        }
      }
    };
    auto add_residual = [&](auto& c) {
#pragma unroll
      for (int j = 0; j < JM; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          c[i][j][0] += __uint_as_float(rr[i][j].x << 16); c[i][j][1] += __uint_as_float(rr[i][j].x & 0xFFFF0000u);
          c[i][j][2] += __uint_as_float(rr[i][j].y << 16); c[i][j][3] += __uint_as_float(rr[i][j].y & 0xFFFF0000u);
        }
    };

    // ring of NS stages, NS - 1 steps of loads in flight.  Step ks is multiplied from buffer ks % NS while the loads of
    // steps ks+1 .. ks+NS-1 travel; a stage is waited for with a COUNTED vmcnt (the younger stages stay in flight
    // across the barrier: raw s_barrier, never __syncthreads(), which would drain them) and read one barrier later.
    int ibuf = RB0;                                                         // buffer of the next stage to issue
    // PRO: every thread transforms, in place, exactly the 16-byte groups it staged itself (its LDS-DMA destinations): the
    // data is complete for it after its own vmcnt wait, no extra barrier is needed before the pass, and its channels are
    // cc*64 + schunk*8 + 0..7 for every one of its rows -- one scale / shift fetch per step.  The pass for step ks+1 runs
    // right behind the MFMAs of step ks (which keep executing), so the loop still has ONE barrier per step.
    float4 ps0 = make_float4(0, 0, 0, 0), ps1 = ps0, ph0 = ps0, ph1 = ps0;
    auto fetch_ss = [&](int cc) {
      const float* p = a.ss + cc * 64 + schunk * 8;
      ps0 = *reinterpret_cast<const float4*>(p); ps1 = *reinterpret_cast<const float4*>(p + 4);
      ph0 = *reinterpret_cast<const float4*>(p + a.C); ph1 = *reinterpret_cast<const float4*>(p + a.C + 4);
    };
    auto transform3 = [&](int buf, int cc, int i0, int i1) {            // PRO3: vectors [i0, i1) of this lane, constants from LDS
      // every LDS access of the pass through the asm helpers (pf_conv_common.h): an ordinary ds_write / ds_read here makes
      // the compiler drain the LDS-DMA ring (vmcnt(0)) once per step
      const uint32_t pss = lds_addr(ssl + cc * 64 + schunk * 8);
      float4 s0, s1, h0, h1;
      // scheduling experiment: the data vectors of the call travel with the constants -- one exposed LDS round trip instead of two
      if (i1 - i0 == 1 || i1 - i0 == 2) {
        const uint32_t pd = lds_addr(smem + buf * STAGE + srow * 128 + (lane & 7) * 16) + (uint32_t)(i0 * (TS / 8) * 128);
        uint4 v0, v1 = make_uint4(0, 0, 0, 0);
        if (i1 - i0 == 1) lds_read_b128x4_1(pss, pss + (uint32_t)a.C * 4u, pd, s0, s1, h0, h1, v0);
        else lds_read_b128x4_2(pss, pss + (uint32_t)a.C * 4u, pd, pd + (TS / 8) * 128, s0, s1, h0, h1, v0, v1);
        pro.sc[0] = s0.x; pro.sc[1] = s0.y; pro.sc[2] = s0.z; pro.sc[3] = s0.w;
        pro.sc[4] = s1.x; pro.sc[5] = s1.y; pro.sc[6] = s1.z; pro.sc[7] = s1.w;
        pro.sh[0] = h0.x; pro.sh[1] = h0.y; pro.sh[2] = h0.z; pro.sh[3] = h0.w;
        pro.sh[4] = h1.x; pro.sh[5] = h1.y; pro.sh[6] = h1.z; pro.sh[7] = h1.w;
        lds_write_b128(pd, pro_apply(pro, v0));
        if (i1 - i0 == 2) lds_write_b128(pd + (TS / 8) * 128, pro_apply(pro, v1));
        return;
      }
      lds_read_b128x4(pss, pss + (uint32_t)a.C * 4u, s0, s1, h0, h1);
      pro.sc[0] = s0.x; pro.sc[1] = s0.y; pro.sc[2] = s0.z; pro.sc[3] = s0.w;
      pro.sc[4] = s1.x; pro.sc[5] = s1.y; pro.sc[6] = s1.z; pro.sc[7] = s1.w;
      pro.sh[0] = h0.x; pro.sh[1] = h0.y; pro.sh[2] = h0.z; pro.sh[3] = h0.w;
      pro.sh[4] = h1.x; pro.sh[5] = h1.y; pro.sh[6] = h1.z; pro.sh[7] = h1.w;
      const uint32_t pv = lds_addr(smem + buf * STAGE + srow * 128 + (lane & 7) * 16);
      static_assert(!PRO3 || AS == 1 || AS == 2 || AS == 4 || AS == 8, "vector count per lane");
#pragma unroll
      for (int i = 0; i < AS; i += 2) {
        if (i >= i0 && i < i1) {
          if (i + 1 < i1) {
            uint4 v0, v1;
            lds_read_b128x2(pv + i * (TS / 8) * 128, pv + (i + 1) * (TS / 8) * 128, v0, v1);
            lds_write_b128(pv + i * (TS / 8) * 128, pro_apply(pro, v0));
            lds_write_b128(pv + (i + 1) * (TS / 8) * 128, pro_apply(pro, v1));
          } else {
            lds_write_b128(pv + i * (TS / 8) * 128, pro_apply(pro, lds_read_b128(pv + i * (TS / 8) * 128)));
          }
        } else if (i + 1 >= i0 && i + 1 < i1) {
          lds_write_b128(pv + (i + 1) * (TS / 8) * 128, pro_apply(pro, lds_read_b128(pv + (i + 1) * (TS / 8) * 128)));
        }
      }
    };
    auto transform = [&](int buf) {
      const float rs[8] = {ps0.x, ps0.y, ps0.z, ps0.w, ps1.x, ps1.y, ps1.z, ps1.w};
      const float rh[8] = {ph0.x, ph0.y, ph0.z, ph0.w, ph1.x, ph1.y, ph1.z, ph1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) { pro.sc[j] = pro_fold_scale(pro, rs[j]); pro.sh[j] = pro_fold_shift(pro, rh[j]); }
      unsigned char* As = smem + buf * STAGE;
#pragma unroll
      for (int i = 0; i < AS; ++i) {
        uint4* p = reinterpret_cast<uint4*>(As + (i * (TS / 8) + srow) * 128 + (lane & 7) * 16);
        *p = pro_apply(pro, *p);
      }
    };
    {
#pragma unroll
    for (int d = 0; d < NS - 1; ++d)
      if (d < nk) {
        if (d > 0 || !pre) {                                                // XPRE: stage 0 was issued before the previous epilogue
          if (PRO && !PRO3) fetch_ss(s_cc);
          stage(ibuf);
        }
        ibuf = (ibuf + 1 == NS) ? 0 : ibuf + 1;
      }
    PF_IG_STAMP(1);                                                         // prologue stages issued
    if (nk >= NS - 1) wait_vm<(NS - 2) * LPS>(); else wait_vm<0>();
    PF_IG_STAMP(2);                                                         // first stage landed
    if (PRO3) { transform3(RB0, 0, 0, AS); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    else if (PRO) { transform(RB0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    PF_IG_STAMP(3);                                                         // first transform + barrier
    int cbuf = RB0;
    for (int ks = 0; ks < nk; ++ks) {
      const bool more = ks + NS - 1 < nk;
      const int tbuf = ibuf;                                                // buffer the stage issued now lands in
      if (more) {
        if (PRO && !PRO3) fetch_ss(s_cc);
        stage(ibuf);
        ibuf = (ibuf + 1 == NS) ? 0 : ibuf + 1;
      }
      const unsigned char* As = smem + cbuf * STAGE;
      const unsigned char* Bs = As + A_BYTES;
      cbuf = (cbuf + 1 == NS) ? 0 : cbuf + 1;
      const bool tnext = PRO3 && (ks + 1 < nk);                             // PRO3: step ks+1 is transformed beside the MFMAs of ks
      if (PRO3) {                                                           // own part of stage ks+1 (issued one step ago) has landed
        // (waiting for its INPUT slots only here and for the kernel slots -- 2/3 of the bytes, issued behind them -- at the end of
        // the step was measured: no difference, profiles/r03_igemm_timeline_split.txt: the steps are bound by the fill RATE)
        if (more) wait_vm<LPS>(); else if (ks + 1 < nk) wait_vm<0>();        // last step: no stage to wait for -- and the residual vectors may still travel
      }
      // experiment: one k-step earlier (its 16 loads per lane sit in front of a step's MFMAs wherever they are issued; here they
      // have two steps to arrive and the LAST step -- the longest of the timeline -- starts its MFMAs at once)
      // (three-stage prologue kernels only: no stage is issued from step nk - 2 on, so no wait of the ring covers these loads)
      if (RPRE && has_r && ks == ((PRO3 && nk >= 2) ? nk - 2 : nk - 1)) load_residual();
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int coff = (((kk * 4 + q) ^ (l15 & 7)) << 4);
        bf16x8 wf[NI], xf[JM];
#if PF_IG_ABLATE != 1
#if PF_IG_FAKE_FRAGS
#pragma unroll
        for (int i = 0; i < NI; ++i) { uint4 u = make_uint4(lane, i, kk, 0x3f803f80u); asm volatile("" : "+v"(u.x)); wf[i] = *reinterpret_cast<const bf16x8*>(&u); }
#pragma unroll
        for (int j = 0; j < JM; ++j) { uint4 u = make_uint4(lane, j, kk, 0x3f803f80u); asm volatile("" : "+v"(u.x)); xf[j] = *reinterpret_cast<const bf16x8*>(&u); }
        (void)coff;
#else
#pragma unroll
        for (int i = 0; i < NI; ++i)
          wf[i] = *reinterpret_cast<const bf16x8*>(Bs + (wn * WC + i * 16 + l15) * 128 + coff);
#ifdef PF_IG_CODES
        const int coffA = ((((kk * 2 + (q >> 1)) ^ ((l15 >> 2) & 3)) << 4) + (q & 1) * 8);
#pragma unroll
        for (int j = 0; j < JM; ++j)
          xf[j] = ig_codes_to_bf16x8(*reinterpret_cast<const uint2*>(As + (wm * WR + j * 16 + l15) * AROW + coffA));
#else
#pragma unroll
        for (int j = 0; j < JM; ++j)
          xf[j] = *reinterpret_cast<const bf16x8*>(As + (wm * WR + j * 16 + l15) * 128 + coff);
#endif
#endif
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < JM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        // Experiment (tools/gpu/build_ablate.sh -> libig_sgb.so): prescribe the interleave of the block.  Left alone, hipcc keeps ONE
        // kernel-fragment register and runs read -> lgkmcnt(0) -> 4 MFMAs -> read -> ... (an exposed LDS round trip per four
        // MFMAs).  With the in-LDS prologue pass between the two halves (asm statements: two scheduling regions) every half asks
        // for its eight fragments first; without it the two halves are ONE region and the second half's fragments are requested
        // under the first half's MFMAs, into the registers it has finished with.
        if constexpr (NI == 4 && JM == 4 && PRO3) {
          __builtin_amdgcn_sched_group_barrier(0x100, NI + JM, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NI * JM, 0);
        }
        if constexpr ((NI == 4 || NI == 2) && JM == 4 && !PRO3) {
          if (kk == 1) {
            // every fragment of the second half is requested under the MFMAs of the first (one read per two MFMAs): ONE exposed
            // LDS round trip per k-step, at its start behind the barrier; 228 registers (236 as hipcc schedules it)
            __builtin_amdgcn_sched_group_barrier(0x100, NI + JM, 0);
#pragma unroll
            for (int i = 0; i < NI + JM; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, (NI * JM) / (NI + JM), 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NI * JM - (NI + JM) * ((NI * JM) / (NI + JM)) + NI * JM, 0);
          }
        }
#else
        (void)coff; (void)wf; (void)xf;
#endif
        // half of this lane's vectors of step ks+1 behind each half of the MFMAs: the VALU work has matrix work to hide in
        if (tnext) transform3(cbuf, ks + 1, kk * (AS / 2), (kk == 1) ? AS : (AS / 2));
      }
      if (!PRO3) {
        if (more) wait_vm<(NS - 2) * LPS>(); else wait_vm<0>();           // the NEXT step's stage has landed (own part)
      }
      if (PRO && !PRO3 && more) transform(tbuf);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // own fragment reads / prologue writes are done
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#ifdef PF_IG_TIMING
      if (ks < 4) PF_IG_STAMP(4 + ks);                                      // k-steps 0..3
      if (ks == nk - 1) PF_IG_STAMP(8);                                     // last k-step
#endif
    }
    }

#if PF_IG_ABLATE >= 4
    {                                                                       // no epilogue: a store that never executes keeps the matrix work alive
      if (a.M < 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < JM; ++j) reinterpret_cast<uint32_t*>(a.Y)[(i * JM + j) * 64 + lane] = pack_bf16x2(acc[i][j][0] + acc[i][j][1], acc[i][j][2] + acc[i][j][3]);
      }
      continue;
    }
#endif
    // ---- epilogue of one [BM][BN] tile (C staging aliases the stage buffers: all reads of them are complete) ----
    {
    if (has_r) {
      if constexpr (RPRE) add_residual(acc);
      else {
#pragma unroll
        for (int j = 0; j < JM; ++j) {                                      // row block by row block: NI vectors live at a time
          const int m = m0 + wm * WR + j * 16 + l15;
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const int n = n0 + wn * WC + i * 16 + q * 4;
            if (m < a.M && n < a.N) {
              const uint2 r = *reinterpret_cast<const uint2*>(a.R + (int64_t)m * a.N + n);
              acc[i][j][0] += __uint_as_float(r.x << 16); acc[i][j][1] += __uint_as_float(r.x & 0xFFFF0000u);
              acc[i][j][2] += __uint_as_float(r.y << 16); acc[i][j][3] += __uint_as_float(r.y & 0xFFFF0000u);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    PF_IG_STAMP(9);                                                         // residual vectors arrived and added
    pre = false;
    if constexpr (XPRE) {
      // behind the residual add on purpose: the compiler waits for the residual registers with vmcnt(0) when a CONDITIONAL
      // batch of loads was issued after them (it cannot count across the branch)
      __builtin_amdgcn_sched_barrier(0);
      if (tm + a.G < a.tiles_m) {                                           // every ring buffer is free behind the last k-step's barrier
        setup_tile((tm + a.G) * BM);
        stage(RB0);
        pre = true;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef PF_IG_CODES
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < JM; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] *= a.kq;                     // alpha of the activation grid (probe entry point)
#endif
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < JM; ++j) {
        const uint2 v = make_uint2(pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3]));
        if constexpr (XPRE) lds_write_b64(lds_addr(Cs + (wm * WR + j * 16 + l15) * CS_LD + wn * WC + i * 16 + q * 4), v);
        else *reinterpret_cast<uint2*>(Cs + (wm * WR + j * 16 + l15) * CS_LD + wn * WC + i * 16 + q * 4) = v;
      }
    }
    // epilogue passes in groups of <= 8 rows per thread: the side vectors (residual / BN input) of a group are all in
    // flight before the group is processed (and before the LDS hand-off for the first group)
    constexpr int PG = (NP > 8) ? 8 : NP;
    uint4 rres[PG];
    auto load_side = [&](int p0) {
#pragma unroll
      for (int p = 0; p < PG; ++p) {
        const int m = m0 + wrw + (p0 + p) * RPP, n = n0 + wvec * 8;
        rres[p] = make_uint4(0, 0, 0, 0);
        if (m < a.M && n < a.N) {
          int64_t srow = m;
          if constexpr (SUB) {                                              // the BN's input is indexed like the OUTPUT: scattered rows
            const int oimg = m / hw_o, orem = m - oimg * hw_o;
            const int oi = orem / a.Wo, oj = orem - oi * a.Wo;
            srow = ((int64_t)oimg * a.o_H + oi * a.o_sub + a.o_y) * a.o_W + oj * a.o_sub + a.o_x;
          }
          rres[p] = *reinterpret_cast<const uint4*>(side + srow * a.N + n);
        }
      }
    };
    constexpr bool etid = true;                                             // every thread takes part in the row passes
    float bpr[32];                                                          // BWD: this thread's 8 channels of scale | shift | mean | invstd
    if (BWD) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int j = 0; j < 8; ++j) bpr[qq * 8 + j] = bpl[qq * BN + wvec * 8 + j];
    }
    if (side != nullptr && etid) load_side(0);
    PF_IG_STAMP(10);                                                        // next tile's first stage issued, C tile packed and written
    if constexpr (XPRE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    else __syncthreads();
    PF_IG_STAMP(11);
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += PG) {
      // scheduling experiment: all C-tile vectors of the group are requested before the first row is processed (as written below,
      // every pass is its own basic block -- read, wait, statistics, store -- and exposes its LDS round trip)
      u32x4_t cv[8];
      if constexpr (XPRE && PG == 8) {
#pragma unroll
        for (int pp = 0; pp < PG; ++pp) lds_read_b128_nowait(lds_addr(Cs + (wrw + (p0 + pp) * RPP) * CS_LD + wvec * 8), cv[pp]);
        lds_wait_batch8(cv);
      } else {
#pragma unroll
        for (int pp = 0; pp < PG; ++pp) cv[pp] = *reinterpret_cast<const u32x4_t*>(Cs + (wrw + (p0 + pp) * RPP) * CS_LD + wvec * 8);
      }
#pragma unroll
      for (int pp = 0; pp < PG; ++pp) {
        const int p = p0 + pp;
        const int rl = wrw + p * RPP;
        const int m = m0 + rl, n = n0 + wvec * 8;
        if (m < a.M && n < a.N && etid) {
          uint4 c;
          c = make_uint4(cv[pp][0], cv[pp][1], cv[pp][2], cv[pp][3]);
          if (BWD) {
            float f[8], xv[8];
            unpack8(c, f);
            unpack8(rres[pp], xv);
            const float* bp = bpr;                                          // hoisted (below): hipcc re-reads the 8 vectors from LDS in every pass
            constexpr int BPS = 8;
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
          int64_t orow = m;
          if constexpr (SUB) {                                              // scatter of a parity class
            const int oimg = m / hw_o, orem = m - oimg * hw_o;
            const int oi = orem / a.Wo, oj = orem - oi * a.Wo;
            orow = ((int64_t)oimg * a.o_H + oi * a.o_sub + a.o_y) * a.o_W + oj * a.o_sub + a.o_x;
          }
          *reinterpret_cast<uint4*>(a.Y + orow * a.N + n) = c;
        }
      }
      if (side != nullptr && p0 + PG < NP && etid) load_side(p0 + PG);
    }
    PF_IG_STAMP(12);                                                        // row passes: statistics, stores issued
    if constexpr (XPRE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    else __syncthreads();
    PF_IG_STAMP(13);
#ifdef PF_IG_TIMING
    if (tm_tile == 2 && blockIdx.x == gridDim.x / 2 && tid == 0) {
      uint32_t* out = reinterpret_cast<uint32_t*>(const_cast<bf16_t*>(a.zero));
#pragma unroll
      for (int k = 0; k < 14; ++k) out[k] = tmk[k];
      out[14] = (uint32_t)__builtin_readcyclecounter();
    }
#endif
  }

  // ---- per-workgroup statistics -> partial[g][stat][N] (fixed order: deterministic) ---------------------------
  if (a.partial != nullptr) {
    // threads with equal column group: RPP row lanes.  Two-level: registers -> LDS [stat][RPP][BN] in chunks that fit
    constexpr int nstat_max = 4;
    const int nstat = BWD ? 2 : 4;
    __syncthreads();
    for (int stat = 0; stat < nstat; ++stat) {
      const float* v = (stat == 0) ? st_s : (stat == 1 ? st_q : (stat == 2 ? st_mn : st_mx));
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wrw * BN + wvec * 8 + j] = v[j];
      __syncthreads();
      for (int c = tid; c < BN; c += T) {
        float r = red[c];
        for (int rr = 1; rr < RPP; ++rr) {
          const float w = red[rr * BN + c];
          r = (stat < 2) ? (r + w) : (stat == 2 ? fminf(r, w) : fmaxf(r, w));
        }
        if (n0 + c < a.N) a.partial[((int64_t)g * nstat + stat) * a.N + n0 + c] = r;
      }
      __syncthreads();
    }
    (void)nstat_max;
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
struct IgCfg { int bm, bn, slots; bool pro3; };    // slots: resident workgroups on the chip (256 CUs x workgroups per CU)

static bool ig_pro3_enabled() {
  return pf_tuning().igemm_pro3 != 0;                      // PF_IGEMM_PRO3=0: round 2's two-stage prologue kernel (A/B runs)
}

// ONE decision for the launcher and for the statistics-group query (the [G][.][N] partial array is sized from it)
static IgCfg ig_pick(int M, int N, bool pro) {
  if (pro) {
    // prologue variant.  Three stages, 8 wavefronts, one workgroup per CU: 128 x 256 tiles where N allows (the in-LDS
    // prologue pass over the input tile is amortised over 256 output channels), 256 x 128 otherwise.
    if (ig_pro3_enabled()) {
      if (N % 256 == 0) return IgCfg{128, 256, 256, true};
      if (N % 128 == 0) return IgCfg{256, 128, 256, true};
    }
    return IgCfg{128, (N % 128 == 0) ? 128 : 64, 512, false};
  }
  {                                                        // PF_IGEMM_TILE, tuning override: "256x128" | "128x128" | "256x64" | "128x64"
    const int bm = pf_tuning().igemm_tile_bm, bn = pf_tuning().igemm_tile_bn;
    if ((bm == 128 || bm == 256) && (bn == 64 || bn == 128) && (N % bn == 0 || bn == 64))
      return IgCfg{bm, bn, (bm == 256) ? 256 : 512, false};
  }
  // (256 x 256 tiles -- eight wavefronts of 64 x 128 -- measured nothing per step in round 5, profiles/r05_first_call_ab.txt; the last
  // instantiation, reachable through PF_IGEMM_TILE only, was removed in round 6)
  const int bn = (N % 128 == 0) ? 128 : 64;
  // measured on the ResNet-50 shapes at batch 256 (tools/gpu/igemm_bench.py): 128-row tiles with two workgroups per CU
  // (2 LDS stages each) beat 256-row tiles with one workgroup per CU and 3 stages on every shape (e.g. 3x3 C = 256 at
  // 14x14: 85 vs 96 us; 3x3 C = 64 at 56x56: 127 vs 165 us): the second workgroup hides the barrier / DMA waits of the
  // first better than a deeper pipeline does
  return IgCfg{128, bn, 512, false};
}

static int ig_grid(int slots, int tiles_m, int tiles_n, int* G_out) {
  int G = slots / tiles_n;
  G = (G / 8) * 8;
  if (G < 8) G = 8;
  const int need = ((tiles_m + 7) / 8) * 8;
  if (G > need) G = need;
  *G_out = G;
  return G * tiles_n;
}

// taps, C: the window and the input channels of the launch (1, K for the 1x1 products)
int pf_igemm_stats_groups_geom(int M, int N, int pro, int taps, int C) {
  (void)taps; (void)C;
  const IgCfg c = ig_pick(M, N, pro != 0);
  int G;
  ig_grid(c.slots, (M + c.bm - 1) / c.bm, (N + c.bn - 1) / c.bn, &G);
  return G;
}

// (M, N) alone: the 1x1 reading -- what pf_conv1x1_stats_groups_k answers for the products it routes here
int pf_igemm_stats_groups(int M, int N, int pro) { return pf_igemm_stats_groups_geom(M, N, pro, 1, 64); }

/* deprecated for RxS convolutions: use pf_conv2d_stats_groups_geom (the kernel, and with it the row count, depends on the window) */
extern "C" int pf_conv2d_stats_groups(int M, int N) { return pf_igemm_stats_groups(M, N, 0); }

template <int BM, int BN, int WM, int WN, int NS, int MODE, bool SUB = false, bool AFF = false>
static int ig_launch_t(IgArgs& a, int slots, hipStream_t st) {
  constexpr bool BWD = (MODE == IG_BWD), PRO3 = (MODE == IG_PRO && NS == 3);
  constexpr int THREADS = 64 * WM * WN;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const int grid = ig_grid(slots, a.tiles_m, a.tiles_n, &a.G);
  // stage ring and (aliased on it) the C tile; the BWD vectors / the folded prologue constants sit behind whichever is larger
  constexpr size_t ring = NS * (size_t)(BM + BN) * 128, ctile = (size_t)BM * (BN + 8) * 2;
  constexpr size_t base = (ring > ctile ? ring : ctile);
  const size_t lds = base + (BWD ? 4 * BN * 4 : 0) + (PRO3 ? 2 * (size_t)a.C * 4 : 0);
  if (PRO3 && a.C > CV_MAXK) return (int)hipErrorInvalidValue;
  const size_t lds_max = base + (BWD ? 4 * BN * 4 : 0) + (PRO3 ? 2 * (size_t)CV_MAXK * 4 : 0);
  if (int e = pf_require_lds(reinterpret_cast<const void*>(&k_igemm<BM, BN, WM, WN, NS, MODE, SUB, AFF>), lds_max)) return e;
  k_igemm<BM, BN, WM, WN, NS, MODE, SUB, AFF><<<grid, THREADS, lds, st>>>(a);
  PF_LAUNCH_CHECK();
  return 0;
}

// pf_conv3x3_c64.hip: the window-staged kernel for 3x3 / stride 1, 64 -> 64 channels on 56 x 56 maps
bool pf_conv3x3_c64_geom(int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w, int Ho, int Wo);
bool pf_conv3x3_c64_takes(const IgArgs& a);
int pf_conv3x3_c64_stats_groups(int M);
int pf_conv3x3_c64_launch(const IgArgs& a, hipStream_t st);

static int ig_launch(IgArgs& a, hipStream_t st) {
  const bool bwd = a.bx != nullptr, pro = a.ss != nullptr;
  if (pf_conv3x3_c64_takes(a)) return pf_conv3x3_c64_launch(a, st);
  const IgCfg c = ig_pick(a.M, a.N, pro);
  if (a.oss != nullptr) {
    // output affine: the dispatcher's default tiles carry it; any other selection (tile override, two-stage prologue kernel) runs
    // the plain launch and the stand-alone pass in place
    if (bwd) return (int)hipErrorInvalidValue;
    if (pro && c.pro3 && a.th * a.tw == 1)
      return (c.bn == 256) ? ig_launch_t<128, 256, 2, 4, 3, IG_PRO, false, true>(a, c.slots, st)
                           : ig_launch_t<256, 128, 4, 2, 3, IG_PRO, false, true>(a, c.slots, st);
    if (!pro && c.bm == 128 && c.bn == 128) return ig_launch_t<128, 128, 2, 2, 2, IG_PLAIN, false, true>(a, c.slots, st);
    if (!pro && c.bm == 128 && c.bn == 64) return ig_launch_t<128, 64, 2, 2, 2, IG_PLAIN, false, true>(a, c.slots, st);
    const float* oss = a.oss;
    a.oss = nullptr;
    const int r = ig_launch(a, st);
    if (r != 0) return r;
    return pf_bn_act_quant_apply(a.Y, a.Y, PF_BF16, a.M, a.N, oss, a.oact, nullptr, 8, 0, st);
  }
  if (pro) {
    if (a.th * a.tw != 1 || bwd) return (int)hipErrorInvalidValue;
    if (c.pro3) return (c.bn == 256) ? ig_launch_t<128, 256, 2, 4, 3, IG_PRO>(a, c.slots, st)
                                     : ig_launch_t<256, 128, 4, 2, 3, IG_PRO>(a, c.slots, st);
    return (c.bn == 128) ? ig_launch_t<128, 128, 2, 2, 2, IG_PRO>(a, c.slots, st) : ig_launch_t<128, 64, 2, 2, 2, IG_PRO>(a, c.slots, st);
  }
#define PF_IG(BMV, BNV, WMV, WNV, NSV) (bwd ? ig_launch_t<BMV, BNV, WMV, WNV, NSV, IG_BWD>(a, c.slots, st) : ig_launch_t<BMV, BNV, WMV, WNV, NSV, IG_PLAIN>(a, c.slots, st))
  if (c.bm == 256 && c.bn == 128) return PF_IG(256, 128, 4, 2, 3);    // 8 wavefronts, 1 workgroup / CU, 3 stages (144 KiB)
  if (c.bm == 128 && c.bn == 128) return PF_IG(128, 128, 2, 2, 2);    // 4 wavefronts, 2 workgroups / CU, 2 stages each
  if (c.bm == 256 && c.bn == 64) return PF_IG(256, 64, 4, 1, 2);
  return PF_IG(128, 64, 2, 2, 2);
#undef PF_IG
}

// rows of the [G][.][N] statistics array pf_conv2d_fwd writes for THIS convolution (depends on the kernel it is dispatched to)
extern "C" int pf_conv2d_stats_groups_geom(int imgs, int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h,
                                           int pad_w, int Ho, int Wo) {
  if (pf_conv3x3_c64_geom(H, Wd, C, N, th, tw, stride, pad_h, pad_w, Ho, Wo) && (imgs * Ho * Wo) % (Ho * Wo) == 0)
    return pf_conv3x3_c64_stats_groups(imgs * Ho * Wo);
  return pf_igemm_stats_groups_geom(imgs * Ho * Wo, N, 0, th * tw, C);
}

// forward convolution (or any implicit GEMM of that form).  X [img][H][Wd][C], W [N][th][tw][C], Y [img][Ho][Wo][N].
// zero: >= 128 zero bytes in device memory.  R / partial / bn_*: epilogue options as for pf_conv1x1_fwd /
// pf_conv1x1_bwd_data_bnstats (partial: [G][4][N] or, with bn_x, [G][2][N]; G = pf_conv2d_stats_groups(M, N)).
static int conv2d_fwd_launch(const void* X, const void* W, void* Y, const void* zero, const void* R, float* partial,
                             const void* bn_x, const float* bn_scale_shift, const float* bn_mean_invstd, int bn_act,
                             int imgs, int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w,
                             int Ho, int Wo, void* stream, const float* out_scale_shift, int out_act) {
  if (out_scale_shift != nullptr && (R != nullptr || partial != nullptr || bn_x != nullptr)) return (int)hipErrorInvalidValue;
  if (imgs <= 0 || H <= 0 || Wd <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || N <= 0 || (C % 64) || (N % 8) || th < 1 || tw < 1 ||
      stride < 1)
    return (int)hipErrorInvalidValue;
  if (!pf_aligned16(X) || !pf_aligned16(W) || !pf_aligned16(Y) || !pf_aligned16(zero) || zero == nullptr ||
      (R && !pf_aligned16(R)) || (bn_x && !pf_aligned16(bn_x)))
    return (int)hipErrorInvalidValue;
  if (bn_x != nullptr && (R != nullptr || partial == nullptr || bn_scale_shift == nullptr || bn_mean_invstd == nullptr))
    return (int)hipErrorInvalidValue;
  if ((int64_t)imgs * H * Wd * C >= ((int64_t)1 << 30) || (int64_t)N * th * tw * C >= ((int64_t)1 << 30) || th * tw > 32)
    return (int)hipErrorInvalidValue;                     // 31-bit byte offsets and a 32-bit tap mask inside the kernel
  IgArgs a;
  a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.Y = (bf16_t*)Y; a.zero = (const bf16_t*)zero;
  a.R = (const bf16_t*)R; a.partial = partial; a.bx = (const bf16_t*)bn_x; a.bss = bn_scale_shift; a.bmi = bn_mean_invstd;
  a.b_lo = (bn_act == PF_ACT_NONE) ? -INFINITY : 0.0f;
  a.b_hi = (bn_act == PF_ACT_RELU6) ? 6.0f : INFINITY;
  a.ss = nullptr; a.slot = nullptr; a.kq = 255.f; a.act_lo = -INFINITY; a.act_hi = INFINITY;
  a.M = imgs * Ho * Wo; a.N = N; a.C = C; a.th = th; a.tw = tw;
  a.H = H; a.Wd = Wd; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  a.x_bytes = (uint32_t)((int64_t)imgs * H * Wd * C * 2);
  a.w_bytes = (uint32_t)((int64_t)N * th * tw * C * 2);
  a.w_r0 = 0; a.w_rs = 1; a.w_s0 = 0; a.w_ss = 1; a.w_S = tw; a.w_taps_full = th * tw;
  a.o_sub = 0; a.o_y = 0; a.o_x = 0; a.o_H = 0; a.o_W = 0;
  a.oss = out_scale_shift; a.oact = out_act;
  return ig_launch(a, (hipStream_t)stream);
}

extern "C" int pf_conv2d_fwd(const void* X, const void* W, void* Y, const void* zero, const void* R, float* partial,
                             const void* bn_x, const float* bn_scale_shift, const float* bn_mean_invstd, int bn_act,
                             int imgs, int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w,
                             int Ho, int Wo, void* stream) {
  return conv2d_fwd_launch(X, W, Y, zero, R, partial, bn_x, bn_scale_shift, bn_mean_invstd, bn_act, imgs, H, Wd, C, N, th, tw,
                           stride, pad_h, pad_w, Ho, Wo, stream, nullptr, PF_ACT_NONE);
}

#ifdef PF_IG_CODES
// variant library only: X holds one byte per element (codes of the grid x = alpha * c), W / Y bf16 as ever
extern "C" int pf_probe_conv2d_fwd_codes(const void* Xc, float alpha, const void* W, void* Y, const void* zero, int imgs, int H,
                                         int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w, int Ho, int Wo,
                                         void* stream) {
  if ((C % 64) || (N % 64) || !pf_aligned16(Xc) || !pf_aligned16(W) || !pf_aligned16(Y)) return (int)hipErrorInvalidValue;
  IgArgs a;
  a.X = (const bf16_t*)Xc; a.W = (const bf16_t*)W; a.Y = (bf16_t*)Y; a.zero = (const bf16_t*)zero;
  a.R = nullptr; a.partial = nullptr; a.bx = nullptr; a.bss = nullptr; a.bmi = nullptr;
  a.b_lo = -INFINITY; a.b_hi = INFINITY;
  a.ss = nullptr; a.slot = nullptr; a.kq = alpha; a.act_lo = -INFINITY; a.act_hi = INFINITY;
  a.M = imgs * Ho * Wo; a.N = N; a.C = C; a.th = th; a.tw = tw;
  a.H = H; a.Wd = Wd; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  a.x_bytes = (uint32_t)((int64_t)imgs * H * Wd * C);
  a.w_bytes = (uint32_t)((int64_t)N * th * tw * C * 2);
  a.w_r0 = 0; a.w_rs = 1; a.w_s0 = 0; a.w_ss = 1; a.w_S = tw; a.w_taps_full = th * tw;
  a.o_sub = 0; a.o_y = 0; a.o_x = 0; a.o_H = 0; a.o_W = 0;
  a.oss = nullptr; a.oact = PF_ACT_NONE;
  const IgCfg c = ig_pick(a.M, a.N, false);
  if (c.bm == 256 && c.bn == 128) return ig_launch_t<256, 128, 4, 2, 3, IG_PLAIN>(a, c.slots, (hipStream_t)stream);
  if (c.bm == 128 && c.bn == 128) return ig_launch_t<128, 128, 2, 2, 2, IG_PLAIN>(a, c.slots, (hipStream_t)stream);
  if (c.bm == 256 && c.bn == 64) return ig_launch_t<256, 64, 4, 1, 2, IG_PLAIN>(a, c.slots, (hipStream_t)stream);
  return ig_launch_t<128, 64, 2, 2, 2, IG_PLAIN>(a, c.slots, (hipStream_t)stream);
}
#endif

// pf_conv2d_fwd with the CONSUMER's inference-mode BN + activation folded into the row pass of the epilogue (see
// pf_conv1x1_fwd_affine): Y = act_out(out_scale[n] * bf16(conv) + out_shift[n]).  No residual, no statistics.
extern "C" int pf_conv2d_fwd_affine(const void* X, const void* W, void* Y, const void* zero, const float* out_scale_shift,
                                    int out_act, int imgs, int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h,
                                    int pad_w, int Ho, int Wo, void* stream) {
  if (out_scale_shift == nullptr) return (int)hipErrorInvalidValue;
  return conv2d_fwd_launch(X, W, Y, zero, nullptr, nullptr, nullptr, nullptr, nullptr, PF_ACT_NONE, imgs, H, Wd, C, N, th, tw,
                           stride, pad_h, pad_w, Ho, Wo, stream, out_scale_shift, out_act);
}

// Backward-data of a STRIDED convolution by output-parity classes (round 4; until then MIOpen).  Forward: y[ho][wo] = sum_{r,s,c}
// x[ho*st + r - pad_h][wo*st + s - pad_w][c] W[n][r][s][c].  The input pixels h = st*i + a of one class a (per axis) receive
// contributions from the taps r = r1 + t*st only (r1 = (a + pad_h) mod st), from output row ho = i + (a + pad_h - r) / st: per
// class a small STRIDE-1 convolution over dY whose taps are a sub-grid of the flipped / transposed kernel buffer
// Wt[c][R-1-r][S-1-s][n] (VarStore.transposed) -- walked in place (IgArgs.w_*), its rows scattered to the class's pixels
// (IgArgs.o_*).  st*st launches, exactly the flops of the convolution, no zero-filled intermediate, no atomics.
// dY [imgs][Ho][Wo][N], Wt [C][R][S][N], dX [imgs][H][W][C]; H % st == 0 and W % st == 0, N % 64 == 0, C % 8 == 0, R, S >= st.
static int strided_class_groups(int imgs, int H, int Wd, int C, int stride) {
  const int bn = (C % 128 == 0) ? 128 : 64;
  int G;
  ig_grid(512, (imgs * (H / stride) * (Wd / stride) + 127) / 128, (C + bn - 1) / bn, &G);
  return G;
}

// rows of the [G][2][C] array pf_conv2d_bwd_data_strided_bnstats writes: stride * stride classes x the workgroup rows of one class
extern "C" int pf_conv2d_bwd_data_strided_stats_groups(int imgs, int H, int Wd, int C, int stride) {
  if (imgs <= 0 || H <= 0 || Wd <= 0 || C <= 0 || stride < 2 || (H % stride) || (Wd % stride)) return 0;
  return stride * stride * strided_class_groups(imgs, H, Wd, C, stride);
}

static int strided_launch(const void* dY, const void* Wt, void* dX, const void* zero, int imgs, int H, int Wd, int C, int N, int R,
                          int S, int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream, float* partial, const void* bn_x,
                          const float* bss, const float* bmi, int bn_act) {
  if (imgs <= 0 || H <= 0 || Wd <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || N <= 0 || (N % 64) || (C % 8) || stride < 2 || R < stride ||
      S < stride || (H % stride) || (Wd % stride) || pad_h < 0 || pad_w < 0 || R * S > 32)
    return (int)hipErrorInvalidValue;
  if (!pf_aligned16(dY) || !pf_aligned16(Wt) || !pf_aligned16(dX) || zero == nullptr) return (int)hipErrorInvalidValue;
  if ((int64_t)imgs * Ho * Wo * N >= ((int64_t)1 << 30) || (int64_t)C * R * S * N >= ((int64_t)1 << 30)) return (int)hipErrorInvalidValue;
  const int Hc = H / stride, Wc = Wd / stride;                  // pixels per class and axis
  for (int ay = 0; ay < stride; ++ay) {
    for (int ax = 0; ax < stride; ++ax) {
      // per axis: first tap of the class, number of taps, offset of the LAST tap's output row relative to i (= the smallest)
      const int r1 = (ay + pad_h) % stride, s1 = (ax + pad_w) % stride;
      const int th = (R - r1 + stride - 1) / stride, tw = (S - s1 + stride - 1) / stride;
      const int dmin_h = (ay + pad_h - r1) / stride - (th - 1), dmin_w = (ax + pad_w - s1) / stride - (tw - 1);
      IgArgs a;
      a.X = (const bf16_t*)dY; a.W = (const bf16_t*)Wt; a.Y = (bf16_t*)dX; a.zero = (const bf16_t*)zero;
      a.R = nullptr; a.partial = nullptr; a.bx = nullptr; a.bss = nullptr; a.bmi = nullptr; a.b_lo = -INFINITY; a.b_hi = INFINITY;
      if (bn_x != nullptr) {                                     // the BN-backward sums of the BN whose input has dX's shape: one
        a.partial = partial + (int64_t)(ay * stride + ax) * strided_class_groups(imgs, H, Wd, C, stride) * 2 * C;   // slice per class
        a.bx = (const bf16_t*)bn_x; a.bss = bss; a.bmi = bmi;
        a.b_lo = (bn_act == PF_ACT_NONE) ? -INFINITY : 0.0f;
        a.b_hi = (bn_act == PF_ACT_RELU6) ? 6.0f : INFINITY;
      }
      a.ss = nullptr; a.slot = nullptr; a.kq = 255.f; a.act_lo = -INFINITY; a.act_hi = INFINITY;
      a.M = imgs * Hc * Wc; a.N = C; a.C = N; a.th = th; a.tw = tw;
      // the launch's "input image" is dY, its "output grid" the class's pixels; tap u reads dY row i + dmin + u = i*1 + u - pad'
      a.H = Ho; a.Wd = Wo; a.Ho = Hc; a.Wo = Wc; a.stride = 1; a.pad_h = -dmin_h; a.pad_w = -dmin_w;
      a.x_bytes = (uint32_t)((int64_t)imgs * Ho * Wo * N * 2);
      a.w_bytes = (uint32_t)((int64_t)C * R * S * N * 2);
      // ascending offset u <-> descending r <-> ascending flipped row R-1-r: first flipped row (R-1-r1) - (th-1)*stride, step stride
      a.w_r0 = (R - 1 - r1) - (th - 1) * stride; a.w_rs = stride;
      a.w_s0 = (S - 1 - s1) - (tw - 1) * stride; a.w_ss = stride;
      a.w_S = S; a.w_taps_full = R * S;
      a.o_sub = stride; a.o_y = ay; a.o_x = ax; a.o_H = H; a.o_W = Wd; a.oss = nullptr; a.oact = PF_ACT_NONE;
      // (the two plain tile configurations the dispatcher picks for these shapes, with the sub-grid walk compiled in)
      int rc;
      if (bn_x != nullptr)
        rc = (C % 128 == 0) ? ig_launch_t<128, 128, 2, 2, 2, IG_BWD, true>(a, 512, (hipStream_t)stream)
                            : ig_launch_t<128, 64, 2, 2, 2, IG_BWD, true>(a, 512, (hipStream_t)stream);
      else
        rc = (C % 128 == 0) ? ig_launch_t<128, 128, 2, 2, 2, IG_PLAIN, true>(a, 512, (hipStream_t)stream)
                            : ig_launch_t<128, 64, 2, 2, 2, IG_PLAIN, true>(a, 512, (hipStream_t)stream);
      if (rc != 0) return rc;
    }
  }
  return 0;
}

extern "C" int pf_conv2d_bwd_data_strided(const void* dY, const void* Wt, void* dX, const void* zero, int imgs, int H, int Wd,
                                          int C, int N, int R, int S, int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream) {
  return strided_launch(dY, Wt, dX, zero, imgs, H, Wd, C, N, R, S, stride, pad_h, pad_w, Ho, Wo, stream, nullptr, nullptr, nullptr,
                        nullptr, PF_ACT_NONE);
}

// ... with the BN-backward sums {sum dy, sum dy * xhat} of the BN that produced the convolution's input in the epilogues (round 6;
// bn2 in front of the strided 3x3 of a stage's first block): partial [G][2][C], G = pf_conv2d_bwd_data_strided_stats_groups(...),
// bn_x [imgs][H][Wd][C] the BN's input, as pf_conv1x1_bwd_data_bnstats.
extern "C" int pf_conv2d_bwd_data_strided_bnstats(const void* dY, const void* Wt, void* dX, const void* zero, const void* bn_x,
                                                  const float* bn_scale_shift, const float* bn_mean_invstd, int bn_act,
                                                  float* partial, int imgs, int H, int Wd, int C, int N, int R, int S, int stride,
                                                  int pad_h, int pad_w, int Ho, int Wo, void* stream) {
  if (bn_x == nullptr || partial == nullptr || bn_scale_shift == nullptr || bn_mean_invstd == nullptr || !pf_aligned16(bn_x))
    return (int)hipErrorInvalidValue;
  return strided_launch(dY, Wt, dX, zero, imgs, H, Wd, C, N, R, S, stride, pad_h, pad_w, Ho, Wo, stream, partial, bn_x,
                        bn_scale_shift, bn_mean_invstd, bn_act);
}

// 1x1 convolutions through the same kernel (called by pf_conv.hip for the shapes it routes here): plain, backward-data with
// BN-backward sums, or with the producer's BN/act/fake-quant prologue; stride > 1 reads input pixel (ho*stride, wo*stride).
// No tap ever leaves the image, `zero` is unused.
int pf_igemm_conv1x1(const void* X, const void* W, void* Y, const void* R, float* partial, const void* bn_x,
                     const float* bss, const float* bmi, float b_lo, float b_hi, const float* scale_shift,
                     const uint32_t* slot, float kq, float act_lo, float act_hi, int M, int N, int K, int Ho, int Wo,
                     int H, int Wd, int stride, const float* oss, int oact, hipStream_t st, int ymap) {
  int64_t rows_in = M;
  if (stride > 1 && !ymap) rows_in = (int64_t)(M / (Ho * Wo)) * H * Wd;
  if ((K % 64) || rows_in * K >= ((int64_t)1 << 30) || (int64_t)N * K >= ((int64_t)1 << 30)) return -1;
  IgArgs a;
  a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.Y = (bf16_t*)Y; a.zero = (const bf16_t*)X;
#ifdef PF_IG_TIMING
  a.zero = (const bf16_t*)(uintptr_t)strtoull(getenv("PF_IG_TIMING_PTR"), nullptr, 0);
#endif
  a.R = (const bf16_t*)R; a.partial = partial; a.bx = (const bf16_t*)bn_x; a.bss = bss; a.bmi = bmi;
  a.b_lo = b_lo; a.b_hi = b_hi;
  a.ss = scale_shift; a.slot = slot; a.kq = kq; a.act_lo = act_lo; a.act_hi = act_hi;
  a.M = M; a.N = N; a.C = K; a.th = 1; a.tw = 1;
  if (ymap) { a.H = Ho; a.Wd = Wo; a.Ho = Ho; a.Wo = Wo; a.stride = 1; }       // dense input rows; the OUTPUT rows are scattered (below)
  else if (stride > 1) { a.H = H; a.Wd = Wd; a.Ho = Ho; a.Wo = Wo; a.stride = stride; }
  else { a.H = 1; a.Wd = 1; a.Ho = 1; a.Wo = 1; a.stride = 1; }
  a.pad_h = 0; a.pad_w = 0;
  a.x_bytes = (uint32_t)(rows_in * K * 2);
  a.w_bytes = (uint32_t)((int64_t)N * K * 2);
  a.w_r0 = 0; a.w_rs = 1; a.w_s0 = 0; a.w_ss = 1; a.w_S = 1; a.w_taps_full = 1;
  a.o_sub = 0; a.o_y = 0; a.o_x = 0; a.o_H = 0; a.o_W = 0;
  a.oss = oss; a.oact = oact;
  if (ymap) {
    // row (img, i, j) of the dense [Ho x Wo] grid goes to pixel (i * stride, j * stride) of the [H x Wd] image: the scatter of the
    // parity-class launches with class (0, 0); the other pixels keep the zeros the caller wrote
    if ((int64_t)(M / (Ho * Wo)) * H * Wd * N >= ((int64_t)1 << 30)) return -1;
    a.o_sub = stride; a.o_y = 0; a.o_x = 0; a.o_H = H; a.o_W = Wd;
    return (N % 128 == 0) ? ig_launch_t<128, 128, 2, 2, 2, IG_PLAIN, true>(a, 512, st) : ig_launch_t<128, 64, 2, 2, 2, IG_PLAIN, true>(a, 512, st);
  }
  return ig_launch(a, st);
}
