// K12, ping-pong form (round 5): the implicit GEMM of pf_igemm.hip -- same operands, same result layout, same epilogue options for the
// plain and the backward-data launches -- on a main loop in which the matrix pipe of a SIMD is never left waiting for the wavefront that
// feeds it.  Replaces, for the shapes it takes, conv2d_fixed_padding's 3x3 convolutions and the deep 1x1 products
// (utils/external/resnet_model.py:92-103, 257-314) and their Conv2DBackpropInput.
//
// Why.  Round 5's fill-rate table (profiles/r05_fill_bench.txt) measured the LDS-DMA path alone at 33-37 TB/s chip-wide (56-60 B/clk/CU)
// -- twice what the per-tap kernels draw -- so the "fill ceiling" of DESIGN.md section 9 (round 4) does not exist.  What the per-tap
// kernels lose is ISSUE time: every wavefront runs  issue 8 LDS-DMA pieces -> 16 fragment reads -> 32 MFMAs -> wait -> barrier  in
// lockstep with the seven others, and while a wavefront sits in the DMA issue (60-185 cycles per piece) or in the LDS round trip its
// SIMD's matrix pipe idles unless the OTHER wavefront of that SIMD happens to be in its MFMA block: 28-47 % busy, measured.
//
// The schedule.  One workgroup of 8 wavefronts per CU = two groups of four, ONE WAVEFRONT OF EACH GROUP PER SIMD (wavefronts w and w + 4
// share a SIMD: the dispatcher hands wavefronts to SIMDs round-robin).  The groups run the same program half a k-step apart:
//
//      interval        2k            2k+1            2k+2           2k+3
//      group 0     MFMA step k   | load phase   | MFMA step k+1  | load phase  ...
//      group 1     load phase    | MFMA step k  | load phase     | MFMA step k+1 ...
//
// (one workgroup barrier per interval).  A load phase = the 16 fragment reads of the NEXT k-step into registers + this wavefront's
// share of LDS-DMA pieces for a stage two (group 0) or three (group 1) steps ahead; an MFMA phase = 32 back-to-back
// v_mfma_f32_16x16x32_bf16 on registers only, then `s_waitcnt vmcnt(0)` for the pieces the wavefront issued one interval ago (they
// had a whole MFMA phase to land).  In every interval one wavefront of each SIMD multiplies while the other one loads: the pipe is
// busy whenever a load phase is not longer than an MFMA phase (512 cycles for a 64 x 64 wavefront tile; the load phase of four
// wavefronts moves 24 KiB through the LDS-DMA path = ~420 cycles at the measured rate and reads 64 KiB of fragments = 256 cycles).
// Every wavefront has at most ONE batch of LDS-DMA in flight and waits for all of it: hipcc's habit of draining vmcnt(0) in front of
// LDS accesses that might alias a pending LDS-DMA costs nothing here, because its waits land where the schedule waits anyway.
//
// Tile.  [bm pixel rows] x [BN = 128 | 64 channels], k-steps of 64 channels per tap, three LDS stages of (256 + BN) x 128 bytes
// (144 KiB).  bm is a RUN-TIME multiple of 16 (<= 256) chosen per launch so that the row tiles divide evenly over the resident
// workgroups -- ResNet-50 at batch 256 has 3.06 (28 x 28), 1.53 (14 x 14) or 0.77 (7 x 7) fixed 128- / 256-row tiles per slot, i.e. the
// busiest CU decides and a quarter of the chip idles (profiles/r04_piece_model.txt); with bm = 208 the same layers are 3.77, 1.89 and
// 0.95 tiles of 13/16 the height.  The 16-row fragments of a tile are dealt round-robin to the four wavefront rows (fragment f belongs
// to row f % 4), so a short tile shortens every wavefront's MFMA phase alike.
//
// Epilogue.  No workgroup-wide C tile: every wavefront stages ITS 64 x (BN/2) block through a private LDS region (aliased on ring
// buffers 0-1; buffer 2 already receives the next tile's first stage), reads it back as 16-byte row vectors and stores full 128-byte
// row segments.  A lane keeps the same 8 channels in every pass: the per-channel statistics of the consumer BN ({sum, sumsq, min, max}
// of the STORED bf16 values) or the BN-backward sums ({sum dy, sum dy * xhat}) of the tile accumulate in registers that live in the
// epilogue only -- the main loop needs every register for the accumulators and a whole k-step of fragments -- and are folded, per
// tile, over the row lanes (through the wavefront's C region, fixed order) into per-wavefront accumulators behind the ring (8 KiB).
// No barrier inside.
#include "pf_igemm.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

// group of a wavefront: 0 / 1.  Wavefronts w and w + 4 share a SIMD (MI355X_MICROARCH.md: a workgroup's wavefronts go to the SIMDs in a
// cyclic order); tools/gpu/pp_bench.py prints the hardware's SIMD id per wavefront to check it.
#ifdef PP_GROUP_ALT            // tools/gpu/build_variant.sh: the other pairing (wavefronts w and w + 1 in different groups), for the A/B only
#define PP_GROUP(wave) ((wave) & 1)
#else
#define PP_GROUP(wave) ((wave) >> 2)
#endif

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <typename F, int... I> __device__ __forceinline__ void pp_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void pp_static_for(F&& f) { pp_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// -DPP_ABLATE=n (tools/gpu/build_variant.sh, never in libpocketflow_hip.so; results are garbage by design): 1 = no LDS-DMA in the main
// loop, 2 = no fragment reads, 3 = neither (MFMAs, waits and barriers only: what the schedule itself costs), 4 = no LDS-DMA of the INPUT
// operand (kernel pieces and all fragment reads remain: the main loop of a kernel that stages an input window once for all taps).
#ifndef PP_ABLATE
#define PP_ABLATE 0
#endif
// -DPP_TIMING (tools/gpu/pp_timeline.py only, never in libpocketflow_hip.so): every wavefront of the middle workgroup keeps s_memtime
// stamps of the first 12 k-steps of its SECOND tile in LDS behind the ring (8 phases per k-step) and writes them to `a.zero` at the end.
#ifdef PP_TIMING
#define PP_STAMP(ph) do { if (tm_rec && tm_k < 12) { const uint32_t t_ = (uint32_t)__builtin_readcyclecounter(); \
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(tm_base + (uint32_t)((tm_k * 8 + (ph)) * 4)), "v"(t_) : "memory"); } } while (0)
// tile-level stamps (row 12 of the table): 0 tile starts, 1 pipeline filled (first fragments read), 2 main loop done, 3 epilogue done
#define PP_STAMP_T(i) do { if (tm_rec) { const uint32_t t_ = (uint32_t)__builtin_readcyclecounter(); \
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(tm_base + (uint32_t)((96 + (i)) * 4)), "v"(t_) : "memory"); } } while (0)
#else
#define PP_STAMP(ph) do { } while (0)
#define PP_STAMP_T(i) do { } while (0)
#endif

template <int BN, int MODE>
__global__ __launch_bounds__(512) void k_igemm_pp(const IgArgs a) {
  constexpr bool BWD = (MODE == IG_BWD);
  constexpr int T = 512, WN = 2;
  constexpr int WC = BN / WN, NI = WC / 16, JM = 4;             // wavefront tile: up to 64 pixels x WC channels
  constexpr int A_BYTES = 256 * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES, NS = 3, RB0 = 2;
  constexpr int AS = 4, BS = BN / 64;                           // LDS-DMA pieces (8 rows x 128 bytes) per wavefront and stage
  constexpr int CS_LD = WC + 8;                                 // bf16 elements per staged C row (16-byte aligned rows, padded)
  constexpr int CWR = 64 * 144;                                 // private LDS region of a wavefront: its C block, then the statistics fold
  constexpr int VPRW = WC / 8, RPW = 64 / VPRW, NPASS = VPRW;   // 16-byte vectors per staged row, rows per pass, passes
  constexpr int OPL = VPRW * 32 / 64;                           // statistics outputs per lane of the per-tile fold (4 | 2)
  static_assert(64 * CS_LD * 2 <= CWR && 8 * CWR <= 2 * STAGE, "the wavefront regions alias ring buffers 0 and 1 only");
  static_assert(BN == 128 || BN == 64, "column tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NS stages; aliased: wavefront regions, statistics scratch
  float* red = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = PP_GROUP(wave);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, q = lane >> 4;
  const int srow = wave * 8 + (lane >> 3);                      // staging row of this lane inside a 64-row slab
  const int schunk = (lane & 7) ^ ((lane >> 3) & 7);            // source 16-byte group for its LDS position (XOR swizzle)

  const int xcd = blockIdx.x & 7, L = blockIdx.x >> 3;
  const int g = xcd + 8 * (L / a.tiles_n), tn = L % a.tiles_n;
  const int n0 = tn * BN;
  const int cch = a.C >> 6;
  const int taps = a.th * a.tw;
  const int nk = taps * cch;
  const int64_t wrow = (int64_t)taps * a.C;
  const int hw_o = a.Ho * a.Wo;
  const int bm = a.pp_bm;
  const int nf = bm >> 4;                                       // 16-row fragments per tile
  // (scalars on purpose: the branches around the fragment rows and the pieces are s_cmp / s_cbranch, not vector compares)
  const int jmn = __builtin_amdgcn_readfirstlane((nf > wm) ? ((nf - wm + 3) >> 2) : 0);         // fragments of this wavefront row: f = j * 4 + wm < nf

  const pf_rsrc_t rsX = PF_MAKE_RSRC(a.X, a.x_bytes);
  const pf_rsrc_t rsW = PF_MAKE_RSRC(a.W, a.w_bytes);
  constexpr uint32_t OOB = 0x80000000u;
  uint32_t boff[BS];
#pragma unroll
  for (int i = 0; i < BS; ++i) {
    const int n = n0 + i * 64 + srow;
    boff[i] = (n < a.N) ? (uint32_t)(((int64_t)n * wrow + schunk * 8) * 2) : OOB;
  }
  const bool pointwise = a.th == 1 && a.tw == 1 && a.stride == 1 && a.pad_h == 0 && a.pad_w == 0 && a.Ho == a.H && a.Wo == a.Wd;

  // Statistics.  Row passes of the epilogue: lane = (row lane rsub, vector vec) keeps channels wn * WC + vec * 8 + 0..7; its 32 values
  // of a tile are  s[8] | q[8] | mn[8] | mx[8].  The per-tile fold leaves output o = vec_o * 32 + val of the wavefront with lane o / OPL,
  // which carries it over the tiles in OPL registers: val = val0 + t, one TYPE per lane (val0 / 8: 0, 1 = sums, 2 = minimum, 3 = maximum).
  const bool want_stats = a.partial != nullptr;
  const int vec = lane % VPRW, rsub = lane / VPRW;
  const int val0 = (lane * OPL) & 31, vec_o = (lane * OPL) >> 5, vtype = val0 >> 3;
  float racc[OPL];
#pragma unroll
  for (int t = 0; t < OPL; ++t) racc[t] = (vtype < 2) ? 0.f : (vtype == 2 ? INFINITY : -INFINITY);

  // fragment read addresses inside a stage: input rows (wm * 16 + l15) + j * 64, kernel rows wn * WC + l15 + i * 16; the 16-byte group
  // of k half kk is ((kk * 4 + q) ^ (l15 & 7)): the second half is the first one with address bit 6 flipped
  const uint32_t fa0 = lds_addr(smem) + (uint32_t)((wm * 16 + l15) * 128 + ((q ^ (l15 & 7)) << 4));
  const uint32_t fb0 = lds_addr(smem) + (uint32_t)(A_BYTES + (wn * WC + l15) * 128 + ((q ^ (l15 & 7)) << 4));

  uint32_t pbase[AS], pmask[AS];
  int s_r = 0, s_s = 0, s_cc = 0, s_ks = 0, s_tap = 0;
  auto setup_tile = [&](int m0) {
#pragma unroll
    for (int i = 0; i < AS; ++i) {
      const int r = i * 64 + srow;
      const int m = m0 + r;
      pbase[i] = 0; pmask[i] = 0;
      if (r < bm && m < a.M) {
        if (pointwise) { pbase[i] = (uint32_t)(m * a.C + schunk * 8) * 2u; pmask[i] = 1u; }
        else {
          const int img = m / hw_o, rem = m - img * hw_o;
          const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
          const int h0 = ho * a.stride - a.pad_h, w0 = wo * a.stride - a.pad_w;
          pbase[i] = (uint32_t)(((img * a.H + h0) * a.Wd + w0) * a.C + schunk * 8) * 2u;   // modulo 2^32 on purpose
          uint32_t mk = 0;
          for (int r2 = 0; r2 < a.th; ++r2)
            for (int sx = 0; sx < a.tw; ++sx)
              if ((unsigned)(h0 + r2) < (unsigned)a.H && (unsigned)(w0 + sx) < (unsigned)a.Wd) mk |= 1u << (r2 * a.tw + sx);
          pmask[i] = mk;
        }
      }
    }
    s_r = 0; s_s = 0; s_cc = 0; s_ks = 0; s_tap = 0;
  };
  // ---- LDS-DMA: this wavefront's pieces of a stage, in tap / channel order (every wavefront issues every stage exactly once, in
  // order).  stage_begin fixes the wave-uniform offsets, piece(p) issues piece p (0 .. AS-1: input rows, AS ..: kernel rows),
  // stage_end advances the tap / channel counters.  NP = pieces per stage of THIS wavefront (what a counted vmcnt leaves in flight).
  uint32_t st_bofs = 0, st_tapoff = 0, st_woff = 0;
  auto stage_begin = [&](int buf) {
    st_bofs = (uint32_t)(buf * STAGE);
    asm volatile("" : "+s"(st_bofs));                           // opaque: one scalar add per piece instead of three hoisted destination sets
    st_tapoff = (uint32_t)(((s_r * a.Wd + s_s) * a.C + s_cc * 64) * 2);
    st_woff = (uint32_t)(s_ks * 128);
  };
  // (the input pieces of a wavefront are as many as its fragment rows: rows i * 64 + wave * 8 < bm  <=>  (i * 4 + wave / 2) * 16 < bm,
  // checked for every height in tools/tile_schedule.py -- so the piece count is a compile-time constant of the tile body, JN + BS)
  auto piece = [&](auto ptag, auto jtag) {
    constexpr int P = decltype(ptag)::value, JN_ = decltype(jtag)::value;
#if PP_ABLATE == 1 || PP_ABLATE == 3
    if (a.M >= 0) return;
#endif
    if constexpr (P < AS) {
#if PP_ABLATE == 4
      if (a.M >= 0) return;
#endif
      if constexpr (P < JN_) {
        const uint32_t voff = ((pmask[P] >> s_tap) & 1u) ? (pbase[P] + st_tapoff) : OOB;
        PF_BUFFER_LOAD_LDS16(rsX, smem + st_bofs + (P * 64 + wave * 8) * 128, voff, 0);
      }
    } else {
      PF_BUFFER_LOAD_LDS16(rsW, smem + st_bofs + A_BYTES + ((P - AS) * 64 + wave * 8) * 128, boff[P - AS], st_woff);
    }
  };
  auto stage_end = [&]() {
    ++s_ks;
    if (++s_cc == cch) { s_cc = 0; ++s_tap; if (++s_s == a.tw) { s_s = 0; ++s_r; } }
  };
  auto stage = [&](int buf, auto jtag) {                        // a whole stage at once (pipeline fill, cross-tile prefetch)
    stage_begin(buf);
    pp_static_for<AS + BS>([&](auto p) { piece(p, jtag); });
    stage_end();
  };
  auto ring = [](int s) { return (RB0 + s) % NS; };             // buffer of stage s of a tile

#ifdef PP_TIMING
  const uint32_t tm_base = lds_addr(smem + NS * STAGE + wave * 512);
  int tm_tile = -1, tm_k = 0;
  bool tm_rec = false;
  for (int i = lane; i < 128; i += 64) asm volatile("ds_write_b32 %0, %1" : : "v"(tm_base + (uint32_t)(i * 4)), "v"(0u) : "memory");
#endif
  bool pre = false;                                             // stage 0 of this tile is already in flight (issued before the last epilogue)
  // One tile, instantiated per fragment-row count JN of the wavefront (a kernel-lifetime constant of the wavefront): the MFMA phase
  // is ONE straight block of 2 * JN * NI instructions.  Round 5's first version branched around every fragment row inside the phase:
  // 770 cycles for 32 MFMAs instead of ~530 (profiles/r05_pp_timeline_v1.txt) -- every extra issue slot between two MFMAs costs tens
  // of cycles (MI355X_MICROARCH.md).
  auto tile_body = [&](auto jn_tag, int tm) {
    constexpr int JN = decltype(jn_tag)::value;
    constexpr int JA = JN ? JN : 1;
    constexpr int NP = JN + BS;                                 // LDS-DMA instructions of this wavefront per stage
    // s_waitcnt vmcnt(n * NP): everything but the newest n batches of this wavefront has landed (LDS-DMA returns in issue order)
    auto wait_keep = [&](int n) {
      if (n <= 0) wait_vm<0>();
      else if (n == 1) wait_vm<NP>();
      else wait_vm<2 * NP>();
    };
    const int m0 = tm * bm;
    PP_STAMP_T(0);
    if (!pre) setup_tile(m0);

    f32x4 acc[NI][JA];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < JA; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // The fragments of ONE k-step: read in a load phase, multiplied in the next MFMA phase.  The reads are inline assembly with ONE
    // address register per (operand, k half) and immediate offsets per fragment: left to hipcc, the 16 addresses x 3 ring buffers are
    // hoisted out of the loop as 48 loop-invariant registers (and the LDS-DMA destinations as 18 scalars), which spilled 250 registers.
    u32x4_t wf[2][NI], xf[2][JA];
    constexpr int NR = 2 * NI + 2 * JN;                         // fragment reads per k-step
#if PP_ABLATE == 2 || PP_ABLATE == 3
#define PP_RD(dst, addr, off) asm volatile("" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#else
#define PP_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#endif
    // A load phase: the fragment reads of a k-step (READ) and this wavefront's pieces of a later stage (STAGE), INTERLEAVED -- three
    // reads, one piece, ... -- so that the LDS queue and the texture-address queue fill side by side: issued one after the other
    // they cost a wavefront 450 + 580 cycles (same timeline), and a wavefront blocked in front of one queue cannot feed the other.
    auto load_phase = [&](auto rtag, auto stag, int rbuf, int sbuf) {
      constexpr bool READ = decltype(rtag)::value, STG = decltype(stag)::value;
      uint32_t pa0 = 0, pa1 = 0, pb0 = 0, pb1 = 0;
      if constexpr (READ) {
        uint32_t bofs = (uint32_t)(rbuf * STAGE);
        asm volatile("" : "+s"(bofs));
        pa0 = fa0 + bofs; pa1 = pa0 ^ 64u; pb0 = fb0 + bofs; pb1 = pb0 ^ 64u;
      }
      if constexpr (STG) stage_begin(sbuf);
      constexpr int NPC = AS + BS;
      // read n of the k-step: kernel fragments first ((k half, fragment) pairs), then the input fragments
#define PP_READ_N(n)                                                                                             \
      do {                                                                                                         \
        if constexpr (READ && (n) < 2 * NI) {                                                                      \
          if constexpr (((n) & 1) == 0) PP_RD(wf[0][((n) >> 1) % NI], pb0, ((n) >> 1) * 2048);                     \
          else PP_RD(wf[1][((n) >> 1) % NI], pb1, ((n) >> 1) * 2048);                                              \
        } else if constexpr (READ && (n) < NR) {                                                                   \
          if constexpr (((n) & 1) == 0) PP_RD(xf[0][(((n) - 2 * NI) >> 1) % JA], pa0, (((n) - 2 * NI) >> 1) * 8192); \
          else PP_RD(xf[1][(((n) - 2 * NI) >> 1) % JA], pa1, (((n) - 2 * NI) >> 1) * 8192);                        \
        }                                                                                                          \
      } while (0)
#define PP_SLOT(g)                                                                  \
      PP_READ_N(3 * (g)); PP_READ_N(3 * (g) + 1); PP_READ_N(3 * (g) + 2);             \
      if constexpr (STG && (g) < NPC) piece(std::integral_constant<int, (g) < NPC ? (g) : 0>{}, jn_tag)
      PP_SLOT(0); PP_SLOT(1); PP_SLOT(2); PP_SLOT(3); PP_SLOT(4); PP_SLOT(5);
#undef PP_SLOT
#undef PP_READ_N
      static_assert(NR <= 18 && NPC <= 6, "six slots of three reads and one piece");
      if constexpr (STG) stage_end();
    };
    // the wait that makes the fragments valid names every destination register, so that no use of them can be scheduled in front of it
    auto frags_wait = [&]() {
      if constexpr (NI == 4)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]), "+v"(wf[0][3]), "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[1][2]), "+v"(wf[1][3]) : : "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[1][0]), "+v"(wf[1][1]) : : "memory");
      if constexpr (JN == 4) asm volatile("" : "+v"(xf[0][0]), "+v"(xf[0][1]), "+v"(xf[0][2]), "+v"(xf[0][3]), "+v"(xf[1][0]), "+v"(xf[1][1]), "+v"(xf[1][2]), "+v"(xf[1][3]) : : "memory");
      else if constexpr (JN == 3) asm volatile("" : "+v"(xf[0][0]), "+v"(xf[0][1]), "+v"(xf[0][2]), "+v"(xf[1][0]), "+v"(xf[1][1]), "+v"(xf[1][2]) : : "memory");
      else if constexpr (JN == 2) asm volatile("" : "+v"(xf[0][0]), "+v"(xf[0][1]), "+v"(xf[1][0]), "+v"(xf[1][1]) : : "memory");
      else if constexpr (JN == 1) asm volatile("" : "+v"(xf[0][0]), "+v"(xf[1][0]) : : "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&]() {
      if constexpr (JN > 0) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int j = 0; j < JN; ++j)
#pragma unroll
            for (int i = 0; i < NI; ++i)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[kk][i]), __builtin_bit_cast(bf16x8, xf[kk][j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    };
    auto phase_end = [&]() {                                    // one interval of the schedule ends: one workgroup barrier
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};
    // a load phase after step k: read the fragments of step k + 1, issue stage k + 3 -- whichever of the two still exists
    auto load_after = [&](int k) {
      if (k + 3 < nk) load_phase(YES, YES, ring(k + 1), ring(k + 3));
      else if (k + 1 < nk) load_phase(YES, NO, ring(k + 1), 0);
    };

    // ---- pipeline fill: stages 0, 1, 2 are issued at once; stage 0 is waited for (the two newer ones stay in flight) ----
    if (!pre) stage(ring(0), jn_tag);
    if (nk > 1) stage(ring(1), jn_tag);
    if (nk > 2) stage(ring(2), jn_tag);
    wait_keep(nk > 2 ? 2 : nk - 1);
    phase_end();
    load_phase(YES, NO, ring(0), 0);
    frags_wait();
    phase_end();
    PP_STAMP_T(1);
    // Steady state: two workgroup barriers per k-step, one behind every interval.  Stage s is issued three steps ahead by both groups:
    // group 0 in interval 2s - 5, group 1 in interval 2s - 4; it is read in intervals 2s - 1 (group 0) and 2s (group 1), so every
    // wavefront's pieces must have landed before the barrier that ends interval 2s - 2: group 0 waits for them at the end of its MFMA
    // phase 2s - 2 (3 intervals after the issue), group 1 at the end of its load phase 2s - 2 (2 intervals) -- each with a COUNTED
    // vmcnt that leaves its newest batch in flight.  The ring buffer of stage s is the one of stage s - 3, last read in interval 2s - 6.
    // (A schedule with ONE barrier per k-step -- the mid-step barrier only enforces the alternation, no memory ordering needs it -- was
    // measured too: no faster, profiles/r05_pp_timeline_v3.txt: the k-step is bound by the LDS / L1 work, not by barrier skew.)
    if (grp == 0) {
      for (int k = 0; k < nk; ++k) {
#ifdef PP_TIMING
        tm_k = k;
#endif
        PP_STAMP(0);
        compute();                                              // interval 2k
        PP_STAMP(1);
        if (k + 1 < nk) wait_keep(k + 2 < nk ? 1 : 0);          // stage k + 1 has landed (own pieces); stage k + 2 stays in flight
        PP_STAMP(2);
        phase_end();
        PP_STAMP(3);
        PP_STAMP(4);
        load_after(k);                                          // interval 2k + 1
        PP_STAMP(5);
        frags_wait();                                           // the fragment reads are complete: the buffer may be refilled behind the barrier
        PP_STAMP(6);
        phase_end();
        PP_STAMP(7);
      }
    } else {
      if (nk > 1) wait_keep(nk > 2 ? 1 : 0);                    // interval 0: own pieces of stage 1 have landed (group 0 reads it in interval 1)
      phase_end();
      for (int k = 0; k < nk; ++k) {
#ifdef PP_TIMING
        tm_k = k;
#endif
        PP_STAMP(0);
        compute();                                              // interval 2k + 1
        PP_STAMP(1);
        PP_STAMP(2);
        phase_end();
        PP_STAMP(3);
        if (k + 1 < nk) {
          PP_STAMP(4);
          load_after(k);                                        // interval 2k + 2
          PP_STAMP(5);
          if (k + 2 < nk) wait_keep(k + 3 < nk ? 1 : 0);        // own pieces of stage k + 2 have landed (group 0 reads it in interval 2k + 3)
          frags_wait();
          PP_STAMP(6);
          phase_end();
          PP_STAMP(7);
        }
      }
    }
    // every fragment read and every LDS-DMA of the tile is complete here, for both groups: the ring is free
    PP_STAMP_T(2);

    // ---- epilogue ----
    if (!BWD && a.R != nullptr) {                               // residual on the fp32 accumulators: ONE rounding to bf16
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        const int m = m0 + (j * 4 + wm) * 16 + l15;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int n = n0 + wn * WC + i * 16 + q * 4;
          if (m < a.M && n < a.N) {
            const uint2 r = *reinterpret_cast<const uint2*>(a.R + (int64_t)m * a.N + n);
            acc[i][j][0] += __uint_as_float(r.x << 16); acc[i][j][1] += __uint_as_float(r.x & 0xFFFF0000u);
            acc[i][j][2] += __uint_as_float(r.y << 16); acc[i][j][3] += __uint_as_float(r.y & 0xFFFF0000u);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    pre = false;
    auto prefetch = [&]() {                                     // the next tile's first stage (ring buffer 2: not touched by the epilogue)
      if (tm + a.G < a.tiles_m) {
        setup_tile((tm + a.G) * bm);
        stage(ring(0), jn_tag);
        pre = true;
      }
    };
    // plain launches: it travels under the whole epilogue.  Backward-data launches have ordinary loads (the BN input vectors and
    // constants) in flight below, and beside a pending LDS-DMA hipcc waits vmcnt(0) for those -- which would wait for the prefetch
    // too: issued behind their last use instead.
    if (!BWD) prefetch();
    __builtin_amdgcn_sched_barrier(0);
    // (every LDS access of the epilogue through the asm helpers: an ordinary one beside the pending LDS-DMA makes hipcc wait vmcnt(0))
    const uint32_t cw = lds_addr(smem + wave * CWR);
#pragma unroll
    for (int j = 0; j < JN; ++j) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const uint2 v = make_uint2(pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3]));
        lds_write_b64(cw + (uint32_t)(((j * 16 + l15) * CS_LD + i * 16 + q * 4) * 2), v);
      }
    }
    // BWD: the BN input vectors of this lane's rows and the BN constants of its 8 channels (re-fetched per tile: 32 registers that
    // the main loop cannot spare)
    uint4 rres[NPASS];
    float bpr[32];
    if (BWD) {
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int lr = p * RPW + rsub;
        const int m = m0 + ((lr >> 4) * 4 + wm) * 16 + (lr & 15), n = n0 + wn * WC + vec * 8;
        rres[p] = make_uint4(0, 0, 0, 0);
        if ((lr >> 4) < JN && m < a.M && n < a.N) rres[p] = *reinterpret_cast<const uint4*>(a.bx + (int64_t)m * a.N + n);
      }
      const int c0 = n0 + wn * WC + vec * 8;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float* src = ((qq < 2) ? a.bss + qq * a.N : a.bmi + (qq - 2) * a.N) + c0;
        const float4 u0 = *reinterpret_cast<const float4*>(src), u1 = *reinterpret_cast<const float4*>(src + 4);
        bpr[qq * 8 + 0] = u0.x; bpr[qq * 8 + 1] = u0.y; bpr[qq * 8 + 2] = u0.z; bpr[qq * 8 + 3] = u0.w;
        bpr[qq * 8 + 4] = u1.x; bpr[qq * 8 + 5] = u1.y; bpr[qq * 8 + 6] = u1.z; bpr[qq * 8 + 7] = u1.w;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (same wavefront writes and reads: the DS queue is in order; the wait pins the order for the compiler)
    __builtin_amdgcn_sched_barrier(0);
    u32x4_t cv[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) lds_read_b128_nowait(cw + (uint32_t)(((p * RPW + rsub) * CS_LD + vec * 8) * 2), cv[p]);
    if constexpr (NPASS == 8) lds_wait_batch8(cv);
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cv[0]), "+v"(cv[1]), "+v"(cv[2]), "+v"(cv[3]) : : "memory");
    float tv[32];                                               // this lane's statistics of THIS tile: s[8] | q[8] | mn[8] | mx[8]
#pragma unroll
    for (int e = 0; e < 8; ++e) { tv[e] = 0.f; tv[8 + e] = 0.f; tv[16 + e] = INFINITY; tv[24 + e] = -INFINITY; }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int lr = p * RPW + rsub;
      const int j = lr >> 4;                                    // (RPW divides 16: one fragment row per pass, wave-uniform)
      const int m = m0 + (j * 4 + wm) * 16 + (lr & 15), n = n0 + wn * WC + vec * 8;
      if (j < JN && m < a.M && n < a.N) {
        const uint4 c = make_uint4(cv[p][0], cv[p][1], cv[p][2], cv[p][3]);
        if (BWD) {
          float f[8], xv[8];
          unpack8(c, f);
          unpack8(rres[p], xv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float u = fmaf(bpr[e], xv[e], bpr[8 + e]);
            const float dy = (u > a.b_lo && u < a.b_hi) ? f[e] : 0.f;
            tv[e] += dy;
            tv[8 + e] = fmaf(dy, (xv[e] - bpr[16 + e]) * bpr[24 + e], tv[8 + e]);
          }
        } else if (want_stats) {
          float f[8];
          unpack8(c, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            tv[e] += f[e];
            tv[8 + e] = fmaf(f[e], f[e], tv[8 + e]);
            tv[16 + e] = pf_acc_min(tv[16 + e], f[e]);
            tv[24 + e] = pf_acc_max(tv[24 + e], f[e]);
          }
        }
        *reinterpret_cast<uint4*>(a.Y + (int64_t)m * a.N + n) = c;
      }
    }
    if (BWD) { __builtin_amdgcn_sched_barrier(0); prefetch(); }
    // ---- fold the tile's statistics over the row lanes, through the wavefront's own region (its C block has been read back): lane
    // (rsub, vec) writes its values at [lane][32 floats, 144-byte stride], lane o / OPL reads the RPW partial values of its OPL
    // outputs and adds them up in row order -- a fixed order, so the sums are reproducible
    if (want_stats) {
      __builtin_amdgcn_sched_barrier(0);
      constexpr int NV = BWD ? 4 : 8;                           // 16-byte groups that carry values (BWD: sums only)
#pragma unroll
      for (int k = 0; k < NV; ++k)
        lds_write_b128(cw + (uint32_t)(lane * 144 + k * 16),
                       make_uint4(__float_as_uint(tv[4 * k]), __float_as_uint(tv[4 * k + 1]), __float_as_uint(tv[4 * k + 2]), __float_as_uint(tv[4 * k + 3])));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (!BWD || vtype < 2) {
        float part[RPW][OPL];
        if constexpr (OPL == 4) {
          u32x4_t pr[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) lds_read_b128_nowait(cw + (uint32_t)((r * VPRW + vec_o) * 144 + val0 * 4), pr[r]);
          lds_wait_batch8(pr);
#pragma unroll
          for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) part[r][t] = __uint_as_float(pr[r][t]);
        } else {
          u32x2_t pr[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) asm volatile("ds_read_b64 %0, %1" : "=&v"(pr[r]) : "v"(cw + (uint32_t)((r * VPRW + vec_o) * 144 + val0 * 4)) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]), "+v"(pr[4]), "+v"(pr[5]), "+v"(pr[6]), "+v"(pr[7]), "+v"(pr[8]),
                         "+v"(pr[9]), "+v"(pr[10]), "+v"(pr[11]), "+v"(pr[12]), "+v"(pr[13]), "+v"(pr[14]), "+v"(pr[15])
                       :
                       : "memory");
#pragma unroll
          for (int r = 0; r < 16; ++r) { part[r][0] = __uint_as_float(pr[r][0]); part[r][1] = __uint_as_float(pr[r][1]); }
        }
        if (vtype < 2) {
#pragma unroll
          for (int t = 0; t < OPL; ++t) {
            float sum = part[0][t];
#pragma unroll
            for (int r = 1; r < RPW; ++r) sum += part[r][t];
            racc[t] += sum;
          }
        } else if (vtype == 2) {
#pragma unroll
          for (int t = 0; t < OPL; ++t)
#pragma unroll
            for (int r = 0; r < RPW; ++r) racc[t] = fminf(racc[t], part[r][t]);
        } else {
#pragma unroll
          for (int t = 0; t < OPL; ++t)
#pragma unroll
            for (int r = 0; r < RPW; ++r) racc[t] = fmaxf(racc[t], part[r][t]);
        }
      }
    }
    // the wavefront regions are free for the next tile's stages when EVERY wavefront is through with its region
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    phase_end();
    PP_STAMP_T(3);
  };
  for (int tm = g; tm < a.tiles_m; tm += a.G) {
#ifdef PP_TIMING
    ++tm_tile;
    tm_rec = (blockIdx.x == gridDim.x / 2) && (tm_tile == ((a.tiles_m > a.G) ? 1 : 0));
#endif
    if (jmn == 4) tile_body(std::integral_constant<int, 4>{}, tm);
    else if (jmn == 3) tile_body(std::integral_constant<int, 3>{}, tm);
    else if (jmn == 2) tile_body(std::integral_constant<int, 2>{}, tm);
    else if (jmn == 1) tile_body(std::integral_constant<int, 1>{}, tm);
    else tile_body(std::integral_constant<int, 0>{}, tm);
  }

#ifdef PP_TIMING
  if (blockIdx.x == gridDim.x / 2) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t* out = reinterpret_cast<uint32_t*>(const_cast<bf16_t*>(a.zero));
    for (int i = lane; i < 128; i += 64) {
      uint32_t v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(tm_base + (uint32_t)(i * 4)) : "memory");
      out[wave * 128 + i] = v;
    }
  }
#endif
  // ---- per-workgroup statistics -> partial[g][stat][N]: the four wavefront rows of a column, in order ----
  if (want_stats) {
    const int nstat = BWD ? 2 : 4;
    __syncthreads();
    // red[stat][wm][BN]: output o = vec_o * 32 + val of wavefront (wm, wn) is channel wn * WC + vec_o * 8 + val % 8 of statistic val / 8
#pragma unroll
    for (int t = 0; t < OPL; ++t) {
      const int val = val0 + t;
      if (val < nstat * 8) red[((val >> 3) * 4 + wm) * BN + wn * WC + vec_o * 8 + (val & 7)] = racc[t];
    }
    __syncthreads();
    for (int i = tid; i < nstat * BN; i += T) {
      const int stat = i / BN, c = i - stat * BN;
      float r = red[(stat * 4) * BN + c];
      for (int w = 1; w < 4; ++w) {
        const float v = red[(stat * 4 + w) * BN + c];
        r = (stat < 2) ? (r + v) : (stat == 2 ? fminf(r, v) : fmaxf(r, v));
      }
      if (n0 + c < a.N) a.partial[((int64_t)g * nstat + stat) * a.N + n0 + c] = r;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
// Row-tile height and walker count of a launch: G walkers per column tile (a multiple of 8: one XCD per walker residue), as many rounds
// as 256-row tiles would need, and the smallest multiple of 16 rows that covers M in that many rounds.
struct PpPlan { int bn, bm, tiles_m, tiles_n, G, grid; };

static PpPlan pp_plan(int M, int N) {
  PpPlan p;
  p.bn = (N % 128 == 0) ? 128 : 64;
  p.tiles_n = (N + p.bn - 1) / p.bn;
  int G = 256 / p.tiles_n;
  G = (G / 8) * 8;
  if (G < 8) G = 8;
  const int64_t rounds = ((int64_t)M + (int64_t)G * 256 - 1) / ((int64_t)G * 256);
  int64_t rows = ((int64_t)M + G * rounds - 1) / (G * rounds);
  int bm = (int)(((rows + 15) / 16) * 16);
  if (bm > 256) bm = 256;
  if (bm < 16) bm = 16;
  const int force = pf_tuning().igemm_pp_bm;                    // PF_IGEMM_PP_BM: tests / sweeps
  if (force >= 16 && force <= 256 && force % 16 == 0) bm = force;
  p.bm = bm;
  p.tiles_m = (M + bm - 1) / bm;
  const int need = ((p.tiles_m + 7) / 8) * 8;
  if (G > need) G = need;
  p.G = G;
  p.grid = G * p.tiles_n;
  return p;
}

// Does the ping-pong kernel take this plain / backward-data launch?  A function of what the statistics-group query knows as well
// (M, N, taps, C): the [G][.][N] array is sized from the same decision.  PF_IGEMM_PP=0 (DEFAULT): never; =2: every shape it can
// compute (tests); =1: where it measured faster than the per-tap kernels LAYER BY LAYER on the ResNet-50 shapes at batch 256, three
// boxes (profiles/r05_pp_bench_v{1,2,3}.txt): the RxS convolutions with at least 18 k-steps and 128-channel column tiles -- 3x3 at
// 28 x 28 and below, forward and backward-data, stride 1 and 2: -3 ... -10 % -- and NOT the 1x1 products (4-32 k-steps per tile: the
// per-tile cost of one workgroup per CU is not amortised, +0 ... +20 %) nor the 64-channel layers at 56 x 56 (+5 %).
// Why it is not the default: in the STEP that selection is 2.2 % SLOWER (10 054 / 10 039 vs 9 836 / 9 810 images/s, one box,
// profiles/r05_pp_step_ab.txt).  The student's forward pass shares the chip with the teacher branch on a second stream; a kernel
// that holds a whole CU's LDS (144 KiB, one workgroup per CU) leaves no room for the other stream's workgroups beside it, while the
// per-tap kernels (two workgroups of 64 KiB per CU) interleave with them -- the overlap is worth more than the kernel's own gain.
bool pf_igemm_pp_takes(int M, int N, int taps, int C) {
  const int mode = pf_tuning().igemm_pp;
  if (mode == 0 || (N % 64) != 0 || (C % 64) != 0 || pf_tuning().igemm_tile_bm != 0) return false;
  if (mode != 2 && pf_grid_share() < 1000) return false;   // a launch at a reduced share of the chip (the teacher's) stays on the per-tap kernels, two workgroups per CU   // (a PF_IGEMM_TILE override asks for a per-tap kernel)
  if (mode == 2) return true;
  return taps >= 9 && (N % 128) == 0 && (int64_t)taps * C >= 1152 && (int64_t)M * N >= ((int64_t)1 << 21);
}

int pf_igemm_pp_stats_groups(int M, int N) { return pp_plan(M, N).G; }

template <int BN, int MODE>
static int pp_launch_t(IgArgs& a, const PpPlan& p, hipStream_t st) {
#ifdef PP_TIMING
  constexpr size_t lds = 3 * (size_t)(256 + BN) * 128 + 4096;
  a.zero = (const bf16_t*)(uintptr_t)strtoull(getenv("PF_PP_TIMING_PTR"), nullptr, 0);
#else
  constexpr size_t lds = 3 * (size_t)(256 + BN) * 128;           // the ring; wavefront regions and the statistics scratch alias it
#endif
  if (int e = pf_require_lds(reinterpret_cast<const void*>(&k_igemm_pp<BN, MODE>), lds)) return e;
  a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n; a.G = p.G; a.pp_bm = p.bm;
  k_igemm_pp<BN, MODE><<<p.grid, 512, lds, st>>>(a);
  PF_LAUNCH_CHECK();
  return 0;
}

// a: filled by the callers of pf_igemm.hip's ig_launch (plain / backward-data launches, no prologue, no sub-filter walk)
int pf_igemm_pp_launch(IgArgs& a, hipStream_t st) {
  if (a.ss != nullptr || (a.N % 64) != 0 || (a.C % 64) != 0) return (int)hipErrorInvalidValue;
  const PpPlan p = pp_plan(a.M, a.N);
  const bool bwd = a.bx != nullptr;
  if (p.bn == 128) return bwd ? pp_launch_t<128, IG_BWD>(a, p, st) : pp_launch_t<128, IG_PLAIN>(a, p, st);
  return bwd ? pp_launch_t<64, IG_BWD>(a, p, st) : pp_launch_t<64, IG_PLAIN>(a, p, st);
}
