// K12, backward-filter (Conv2DBackpropFilter of the student's convolutions, utils/external/resnet_model.py:92-103,
// :257-314) with the producer's BN + ReLU + fake-quant re-applied to the input operand (K13 / K4 prologue):
//
//     dW[n][tap][c] = sum_m dY[m][n] * Q(X)[pix(m, tap)][c]          m = (img, ho, wo)
//
// The contraction runs over PIXELS, the slow index of both NHWC operands, so both MFMA operands are "transposed".
// gfx950 reads a [4 pixels][16 channels] bf16 block out of LDS transposed in one instruction (ds_read_b64_tr_b16:
// lane i of a 16-lane group supplies the i-th 8-byte piece of the block and receives column i); this kernel is
// built around it:
//   * tiles go from global memory straight into LDS (global_load_lds, no staging registers, no 2-byte scatter) in a
//     block layout [8 pixels][16 channels] = 256 contiguous bytes per block, so that the 32 lanes the LDS services
//     together read one full 256-byte line: bank-conflict free.  The order in which pixels map to MFMA k-slots is a free
//     permutation (a contraction index), chosen for exactly that;
//   * every WAVEFRONT owns its own pixel stream and accumulates the whole [BN x 64] output tile of the workgroup: each
//     element of dY and X is read by exactly one wavefront (the prologue runs once per element, on one channel per
//     lane -- scale / shift live in two registers), nothing is shared in the main loop, there is NO barrier in it,
//     and each wavefront keeps three stages (24-36 KiB) of LDS-DMA in flight behind counted vmcnt waits;
//   * the four wavefronts are combined through LDS in a fixed order, pixel splits through the staged reduction of
//     pf_conv.hip (k_wrw_reduce): bit-reproducible, no float atomics.
// The (n, k) tiles of one pixel split occupy consecutive workgroups of one XCD, so the rows they all read are fetched
// from HBM once and shared through that XCD's L2.
#include "pf_conv_common.h"
#include <stdlib.h>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

// source of out-of-range rows / padding taps: the products must vanish, so these are real zeros (not "any readable bytes")
__device__ __attribute__((aligned(16))) const uint32_t pf_wrw_zero_page[64] = {0};

struct WrwArgs {
  const bf16_t* dY;     // [M][N]
  const bf16_t* X;      // [rows_in][C]
  float* slabs;         // [S][N][taps*C]
  const float* ss;      // prologue scale | shift [2][C] or null
  const uint32_t* slot;
  float kq, act_lo, act_hi;
  int M, N, C;
  int th, tw, H, Wd, Ho, Wo, stride, pad_h, pad_w;
  int tiles_n, tiles, rows_per_split;
};

template <int N> __device__ __forceinline__ void wrw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BN: output channels of the tile (64 | 128); PRO: prologue on X; MAP: taps / strides (input row != output row)
template <int BN, bool PRO, bool MAP>
__global__ __launch_bounds__(256) void k_wrw_tr(const WrwArgs a) {
  constexpr int NB = BN / 16;                       // 16-channel blocks of the dY tile
  constexpr int NS = 3;                             // LDS stages per wavefront
  constexpr int DY_BYTES = 4 * NB * 256, X_BYTES = 4 * 4 * 256, STAGE = DY_BYTES + X_BYTES;   // 32 pixels per stage
  constexpr int LPS = NB + 4;                       // LDS-DMA instructions per lane and stage
  constexpr int NBLK = NB * 4;                      // 16x16 accumulator blocks
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  unsigned char* ring = smem + wave * (NS * STAGE);
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(pf_wrw_zero_page);

  // XCD-aware order (see k_conv1x1_wrw): the tiles of one pixel split are consecutive on one XCD
  const int nwg = gridDim.x;
  int wg = blockIdx.x;
  {
    const int xcd = wg & 7, idx = wg >> 3;
    const int qq = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
  }
  const int tile = wg % a.tiles, split = wg / a.tiles;
  const int tn = tile % a.tiles_n, tk = tile / a.tiles_n;
  const int n0 = tn * BN;
  const int cch = a.C >> 6;
  const int tap = tk / cch, c0 = (tk - tap * cch) * 64;
  const int tap_r = tap / a.tw, tap_s = tap - tap_r * a.tw;
  const int ktot = a.th * a.tw * a.C;
  const int mbeg = split * a.rows_per_split;
  const int mend = (mbeg + a.rows_per_split < a.M) ? (mbeg + a.rows_per_split) : a.M;
  const int nsteps = (mend > mbeg) ? (mend - mbeg + 31) / 32 : 0;
  const int my_steps = (nsteps > wave) ? (nsteps - wave + 3) / 4 : 0;      // wavefront w takes steps w, w+4, ...

  // prologue constants: this lane's channels are c0 + j*16 + l15 in every X fragment
  float psc[4], psh[4];
  Pro pro;
  pro_init(pro, a.act_lo, a.act_hi, a.kq, PRO ? a.slot : nullptr);
  if (PRO) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      psc[j] = pro_fold_scale(pro, a.ss[c0 + j * 16 + l15]);
      psh[j] = pro_fold_shift(pro, a.ss[a.C + c0 + j * 16 + l15]);
    }
  }

  // staging roles of this lane inside one LDS-DMA instruction: block b (of 4), pixel row (of 8), 16-byte half
  const int sb = lane >> 4, srow = (lane & 15) >> 1, sch = lane & 1;
  const int hw_o = a.Ho * a.Wo;
  auto stage = [&](int step, int buf) {
    unsigned char* dst = ring + buf * STAGE;
    const int mb = mbeg + step * 32;
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
      const int m = mb + pg * 8 + srow;
      const bool ok = m < mend;
      // dY: NB/4 instructions of 4 channel blocks each
#pragma unroll
      for (int g4 = 0; g4 < NB / 4; ++g4) {
        const int n = n0 + (g4 * 4 + sb) * 16 + sch * 8;
        const bf16_t* src = (ok && n < a.N) ? (a.dY + (int64_t)m * a.N + n) : (zero + sch * 8);
        __builtin_amdgcn_global_load_lds(src, LDS_PTR(dst + (pg * NB + g4 * 4) * 256), 16, 0, 0);
      }
      // X: one instruction (4 channel blocks = the 64-channel step)
      bool okx = ok;
      int64_t row = m;
      if (MAP && ok) {
        const int img = m / hw_o, rem = m - img * hw_o;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        const int hi = ho * a.stride + tap_r - a.pad_h, wi = wo * a.stride + tap_s - a.pad_w;
        okx = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.Wd;
        row = ((int64_t)img * a.H + hi) * a.Wd + wi;
      }
      const bf16_t* srcx = okx ? (a.X + row * a.C + c0 + sb * 16 + sch * 8) : (zero + sch * 8);
      __builtin_amdgcn_global_load_lds(srcx, LDS_PTR(dst + DY_BYTES + pg * 4 * 256), 16, 0, 0);
    }
  };

  f32x4 acc[NB][4];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed-read address of this lane inside a block pair: k-slot group q -> pixel block (q >> 1) + 2h, half q & 1
  const int tr_off = (q & 1) * 128 + (l15 >> 2) * 32 + (l15 & 3) * 8;
  const int pg_lo = q >> 1;                                               // h = 0: blocks 0 / 1, h = 1: blocks 2 / 3

  if (my_steps > 0) stage(wave, 0);
  if (my_steps > 1) stage(wave + 4, 1);
  int ibuf = 2, cbuf = 0;
  for (int t = 0; t < my_steps; ++t) {
    if (t + 2 < my_steps) {
      stage(wave + (t + 2) * 4, ibuf);
      ibuf = (ibuf == NS - 1) ? 0 : ibuf + 1;
      wrw_wait_vm<2 * LPS>();
    } else if (t + 1 < my_steps) {
      wrw_wait_vm<LPS>();
    } else {
      wrw_wait_vm<0>();
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* sbase = ring + cbuf * STAGE;
    cbuf = (cbuf == NS - 1) ? 0 : cbuf + 1;
    // X fragments (operand B: columns = channels), prologue on the registers
    bf16x8 xf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) v4s*)(sbase + DY_BYTES + (pg_lo * 4 + j) * 256 + tr_off));
      const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) v4s*)(sbase + DY_BYTES + ((pg_lo + 2) * 4 + j) * 256 + tr_off));
      uint4 u;
      u.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
      u.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
      u.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
      u.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
      if (PRO) {
        float f[8];
        unpack8(u, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = pro_point(pro, psc[j], psh[j], f[e]);
        u = pack8(f);
      }
      xf[j] = *reinterpret_cast<const bf16x8*>(&u);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) v4s*)(sbase + (pg_lo * NB + i) * 256 + tr_off));
      const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) v4s*)(sbase + ((pg_lo + 2) * NB + i) * 256 + tr_off));
      v8s d8;
      d8[0] = lo[0]; d8[1] = lo[1]; d8[2] = lo[2]; d8[3] = lo[3];
      d8[4] = hi[0]; d8[5] = hi[1]; d8[6] = hi[2]; d8[7] = hi[3];
      const bf16x8 df = *reinterpret_cast<const bf16x8*>(&d8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, xf[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- the four wavefronts -> one tile (fixed order), then the split's slab --------------------------------
  __syncthreads();                                   // every ring is idle (each wavefront ended on vmcnt(0))
  float4* red = reinterpret_cast<float4*>(smem);     // [4 wavefronts][NBLK][64 lanes]
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      red[(wave * NBLK + i * 4 + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  __syncthreads();
  float* out = a.slabs + (int64_t)split * a.N * ktot;
  const int kcol = tap * a.C + c0;
#pragma unroll
  for (int b = 0; b < NBLK / 4; ++b) {
    const int blk = wave * (NBLK / 4) + b;
    const int i = blk >> 2, j = blk & 3;
    float4 s = red[(0 * NBLK + blk) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 v = red[(w * NBLK + blk) * 64 + lane];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int k = kcol + j * 16 + l15;
    const int n = n0 + i * 16 + q * 4;
    const float e[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < a.N) out[(int64_t)(n + r) * ktot + k] = e[r];
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
// Measured (tools/gpu/wrw_bench.py, ResNet-50 shapes, batch 256): correct and deterministic, but 15-25 % SLOWER than the
// scatter kernel of pf_conv.hip on the 1x1 layers (99 vs 84 us at 14x14 1024->256) and 2.7x slower than MIOpen on the 3x3
// ones (336 vs 121 us): with every wavefront accumulating the whole [128 x 64] tile the kernel moves 12 KiB of LDS-DMA
// per 32 MFMAs (43 flop / byte) through one workgroup of 4 wavefronts per CU -- load-bandwidth bound per CU.  The 1x1
// dispatcher therefore uses it only on request (PF_WRW_TR=1); pf_conv2d_wrw (the RxS entry point, which has no
// other implementation here) always runs it.  The fix is a shared 256 x 256 output tile.
static bool wrw_tr_1x1_enabled() {
  return pf_tuning().wrw_tr != 0;                          // PF_WRW_TR
}

int pf_wrw_tr_splits(int M, int N, int C, int taps);
// the 1x1 dispatcher's gate (pf_conv.hip): > 0 -> this many pixel splits on a kernel of this file
int pf_wrw2_splits(int M, int N, int C, int taps);
int pf_wrw_tr_splits_1x1(int M, int N, int C) {
  const int s2 = pf_wrw2_splits(M, N, C, 1);
  if (s2 > 0) return s2;
  return wrw_tr_1x1_enabled() ? pf_wrw_tr_splits(M, N, C, 1) : 0;
}

static int wrw_tr_bn(int N) { return (N % 128 == 0) ? 128 : 64; }

// pixel splits of the transposed-read kernel (0: the kernel does not apply)
int pf_wrw_tr_splits(int M, int N, int C, int taps) {
  if ((C % 64) || (N % 64) || M < 2048) return 0;
  const int bn = wrw_tr_bn(N);
  const int tiles = (N / bn) * (taps * C / 64);
  int S = (256 + tiles - 1) / tiles;                      // ~ one workgroup (4 wavefronts, <= 144 KiB LDS) per CU
  const int maxS = (M + 511) / 512;                       // >= 4 steps of 32 pixels per wavefront
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  int rows = (M + S - 1) / S;
  rows = ((rows + 127) / 128) * 128;
  return (M + rows - 1) / rows;
}

template <int BN, bool PRO, bool MAP>
static int wrw_tr_launch_t(const WrwArgs& a, int grid, hipStream_t st) {
  constexpr int NB = BN / 16;
  const size_t ring = 4 * (size_t)3 * ((4 * NB + 16) * 256);
  const size_t red = (size_t)4 * NB * 4 * 64 * 16;
  const size_t lds = ring > red ? ring : red;
  if (int e = pf_require_lds(reinterpret_cast<const void*>(&k_wrw_tr<BN, PRO, MAP>), lds)) return e;
  k_wrw_tr<BN, PRO, MAP><<<grid, 256, lds, st>>>(a);
  PF_LAUNCH_CHECK();
  return 0;
}

// slabs: [S][N][taps*C] floats; returns -1 when the kernel does not apply, else a hipError_t
int pf_wrw_tr_launch(const void* dY, const void* X, float* slabs, const float* scale_shift, int act,
                     const uint32_t* slot, int bits, int M, int N, int C, int th, int tw, int H, int Wd, int Ho, int Wo,
                     int stride, int pad_h, int pad_w, int S, hipStream_t st) {
  if (S <= 0) return -1;
  WrwArgs a;
  a.dY = (const bf16_t*)dY; a.X = (const bf16_t*)X; a.slabs = slabs;
  a.ss = scale_shift; a.slot = slot;
  a.kq = uq_k_of_bits(slot ? bits : 8);
  a.act_lo = (act == PF_ACT_NONE) ? -INFINITY : 0.0f;
  a.act_hi = (act == PF_ACT_RELU6) ? 6.0f : INFINITY;
  a.M = M; a.N = N; a.C = C; a.th = th; a.tw = tw; a.H = H; a.Wd = Wd; a.Ho = Ho; a.Wo = Wo;
  a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  const int bn = wrw_tr_bn(N);
  a.tiles_n = N / bn;
  a.tiles = a.tiles_n * (th * tw * C / 64);
  int rows = (M + S - 1) / S;
  rows = ((rows + 127) / 128) * 128;
  a.rows_per_split = rows;
  const int grid = a.tiles * S;
  const bool pro = scale_shift != nullptr;
  const bool map = stride != 1 || th * tw > 1;
  if (pro && th * tw > 1) return -1;                      // padding taps need Q = 0, not Q(0): materialised inputs only
#define PF_WT(BNV)                                                                            \
  do {                                                                                        \
    if (pro) return map ? wrw_tr_launch_t<BNV, true, true>(a, grid, st) : wrw_tr_launch_t<BNV, true, false>(a, grid, st);   \
    return map ? wrw_tr_launch_t<BNV, false, true>(a, grid, st) : wrw_tr_launch_t<BNV, false, false>(a, grid, st);          \
  } while (0)
  if (bn == 128) PF_WT(128);
  PF_WT(64);
#undef PF_WT
}

// =====================================================================================================================
// Second generation: SHARED output tile.  The wave-private kernel above moves 12 KiB of LDS-DMA per 32 MFMAs (43 flop per
// byte) through 4 wavefronts per CU and is load-bandwidth bound.  Here a workgroup of 4 / 8 wavefronts shares one staged
// [32 pixels][TN + TK channels] tile per step and every wavefront owns a disjoint [WTN x WTK] block of the [TN x TK] output
// tile (256 x 128: 85 flop per byte, no cross-wavefront reduction at all).  Same block layout and transposed reads as
// above; staging through buffer_load ... lds (out-of-range rows / padding taps read zeros from the bounds check); two LDS
// stages, ONE barrier per step; with the prologue, every thread transforms in place exactly the 16-byte groups of X it
// staged itself (8 consecutive channels of one pixel: its scale / shift live in registers for the whole launch), right
// behind the MFMAs of the previous step.
// =====================================================================================================================
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t pf_wrsrc_t;
#define PF_W_MAKE_RSRC(p, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(p), (short)0, (int)(bytes), 0x00020000)
#define PF_W_BUFFER_LOAD_LDS16(rs, lds, voff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, voff, 0, 0, 0)
#else
typedef int pf_wrsrc_t;                               // host pass: the kernel body only has to parse (see pf_igemm.hip)
#define PF_W_MAKE_RSRC(p, bytes) 0
#define PF_W_BUFFER_LOAD_LDS16(rs, lds, voff) ((void)(rs), (void)(lds), (void)(voff))
#endif

// Ablation builds for tools/gpu/wrw_ablate.py ONLY (never in libpocketflow_hip.so): PF_W2_ABLATE bit 0 drops the fragment reads
// and MFMAs of k_wrw2, bit 1 the LDS-DMA, bit 2 the slab stores of the epilogue.  Results are garbage by design.
#ifndef PF_W2_ABLATE
#define PF_W2_ABLATE 0
#endif
// -DPF_W2_TIMING (tools/gpu/wrw_timeline.py only): lane 0 of every wavefront of the middle workgroup records s_memtime at 7
// points of steps 4..11 into the first words of the slab workspace; the kernel returns before its epilogue then.
// (plain global stores: they ride on vmcnt like the LDS-DMA, which can only lengthen the counted wait a little)
#ifdef PF_W2_TIMING
#define PF_W2_STAMP(k) do { if (blockIdx.x == (unsigned)tm_blk && t >= 4 && t < 12 && lane == 0) \
    tm_out[(t - 4) * 8 + (k)] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
#define PF_W2_STAMP(k) do { } while (0)
#endif

// MFMA fragments (8 consecutive pixels of one channel each: two transposing reads, `lo` at p + off and `hi` at
// p + off + HI) through INLINE ASM.  Reason: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first
// __builtin_amdgcn_ds_read_tr16_b64 it meets while an LDS-DMA is pending (it treats the DMA as a store the read may alias):
// in k_wrw2 that wait sat right behind the issue of the NEXT stage's loads, i.e. every step waited for the prefetch it had
// just started -- no overlap of loads and matrix work at all.  The asm reads are invisible to that pass; the statement
// carries its own lgkmcnt(0), so the results are valid when it ends.  The caller guarantees the stage has landed (own
// counted vmcnt + barrier).
__device__ __forceinline__ bf16x8 tr_join(const v4s& lo, const v4s& hi) {
  v8s x8;
  x8[0] = lo[0]; x8[1] = lo[1]; x8[2] = lo[2]; x8[3] = lo[3];
  x8[4] = hi[0]; x8[5] = hi[1]; x8[6] = hi[2]; x8[7] = hi[3];
  return *reinterpret_cast<const bf16x8*>(&x8);
}
template <int HI>
__device__ __forceinline__ void tr_read_frags(uint32_t p, bf16x8 (&f)[4]) {
  v4s l0, l1, l2, l3, h0, h1, h2, h3;
  asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %4, %8 offset:%9\n\t"
               "ds_read_b64_tr_b16 %1, %8 offset:256\n\tds_read_b64_tr_b16 %5, %8 offset:%10\n\t"
               "ds_read_b64_tr_b16 %2, %8 offset:512\n\tds_read_b64_tr_b16 %6, %8 offset:%11\n\t"
               "ds_read_b64_tr_b16 %3, %8 offset:768\n\tds_read_b64_tr_b16 %7, %8 offset:%12\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3)
               : "v"(p), "n"(HI), "n"(HI + 256), "n"(HI + 512), "n"(HI + 768)
               : "memory");
  f[0] = tr_join(l0, h0); f[1] = tr_join(l1, h1); f[2] = tr_join(l2, h2); f[3] = tr_join(l3, h3);
}
template <int HI>
__device__ __forceinline__ void tr_read_frags(uint32_t p, bf16x8 (&f)[2]) {
  v4s l0, l1, h0, h1;
  asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %2, %4 offset:%5\n\t"
               "ds_read_b64_tr_b16 %1, %4 offset:256\n\tds_read_b64_tr_b16 %3, %4 offset:%6\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(l0), "=&v"(l1), "=&v"(h0), "=&v"(h1)
               : "v"(p), "n"(HI), "n"(HI + 256)
               : "memory");
  f[0] = tr_join(l0, h0); f[1] = tr_join(l1, h1);
}

// BOTH operands' fragments behind one wait (round 4; measured with the other scheduling changes, +2.2 % per step) -- the two calls above expose
// two LDS round trips per k-step of ~16 MFMAs
template <int HIX, int HID>
__device__ __forceinline__ void tr_read_frags2(uint32_t px, uint32_t pd, bf16x8 (&xf)[4], bf16x8 (&df)[4]) {
  v4s a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3, d0, d1, d2, d3;
  asm volatile("ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %4, %16 offset:%18\n\t"
               "ds_read_b64_tr_b16 %1, %16 offset:256\n\tds_read_b64_tr_b16 %5, %16 offset:%19\n\t"
               "ds_read_b64_tr_b16 %2, %16 offset:512\n\tds_read_b64_tr_b16 %6, %16 offset:%20\n\t"
               "ds_read_b64_tr_b16 %3, %16 offset:768\n\tds_read_b64_tr_b16 %7, %16 offset:%21\n\t"
               "ds_read_b64_tr_b16 %8, %17\n\tds_read_b64_tr_b16 %12, %17 offset:%22\n\t"
               "ds_read_b64_tr_b16 %9, %17 offset:256\n\tds_read_b64_tr_b16 %13, %17 offset:%23\n\t"
               "ds_read_b64_tr_b16 %10, %17 offset:512\n\tds_read_b64_tr_b16 %14, %17 offset:%24\n\t"
               "ds_read_b64_tr_b16 %11, %17 offset:768\n\tds_read_b64_tr_b16 %15, %17 offset:%25\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3),
                 "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
               : "v"(px), "v"(pd), "n"(HIX), "n"(HIX + 256), "n"(HIX + 512), "n"(HIX + 768), "n"(HID), "n"(HID + 256), "n"(HID + 512),
                 "n"(HID + 768)
               : "memory");
  xf[0] = tr_join(a0, b0); xf[1] = tr_join(a1, b1); xf[2] = tr_join(a2, b2); xf[3] = tr_join(a3, b3);
  df[0] = tr_join(c0, d0); df[1] = tr_join(c1, d1); df[2] = tr_join(c2, d2); df[3] = tr_join(c3, d3);
}

struct Wrw2Args {
  const bf16_t* dY;
  const bf16_t* X;
  float* slabs;
  const float* ss;
  const uint32_t* slot;
  float kq, act_lo, act_hi;
  uint32_t dy_bytes, x_bytes;
  int M, N, C;
  int th, tw, H, Wd, Ho, Wo, stride, pad_h, pad_w;
  int tiles_n, tiles, rows_per_split;
};

// MAPM: how an output pixel m maps to the X row it multiplies -- 0: identity (1x1, stride 1); 1: "same-size shift": an RxS
// tap of a stride-1 convolution whose output has the input's size: row = m + (r - pad_h) * Wd + (s - pad_w) inside the image,
// nothing outside (all address arithmetic incremental: one add per LDS-DMA and step); 2: general (strided) map.
template <int TN, int TK, int WTN, int WTK, bool PRO, int MAPM>
__global__ __launch_bounds__(64 * (TN / WTN) * (TK / WTK)) void k_wrw2(const Wrw2Args a) {
  constexpr int WN = TN / WTN, WK = TK / WTK, NWAVE = WN * WK;
  constexpr int NI = WTN / 16, NJ = WTK / 16;        // 16x16 accumulator blocks of one wavefront
  constexpr int NBN = TN / 16, NBK = TK / 16;        // 16-channel blocks of the staged tiles
  constexpr int DY_BYTES = 4 * NBN * 256, X_BYTES = 4 * NBK * 256, STAGE = DY_BYTES + X_BYTES;
  constexpr int IDY = 4 * (TN / 64), IX = 4 * (TK / 64);                     // LDS-DMA instructions per stage
  static_assert(IDY % NWAVE == 0 || IDY < NWAVE, "dY instructions must split evenly over the wavefronts (slot kinds are static)");
  constexpr int KDY = (IDY + NWAVE - 1) / NWAVE;     // slots 0..KDY-1 of every wavefront stage dY, the rest stage X (IDY < NWAVE -- the
                                                     // 64 x 256 tile of round 6 -- : wavefronts IDY.. load zeros into the sink, like X)
  constexpr int XS = (IX + NWAVE - 1) / NWAVE;       // X slots per wavefront (slot x is live iff wave + x*NWAVE < IX)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int wn = wave / WK, wk = wave % WK;

  const int nwg = gridDim.x;
  int wg = blockIdx.x;
  {
    const int xcd = wg & 7, idx = wg >> 3;
    const int qq = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
  }
  const int tile = wg % a.tiles, split = wg / a.tiles;
  const int tn = tile % a.tiles_n, tk = tile / a.tiles_n;
  const int n0 = tn * TN;
  const int kt = a.C / TK;                            // channel tiles per tap
  const int tap = tk / kt, c0 = (tk - tap * kt) * TK;
  const int tap_r = tap / a.tw, tap_s = tap - tap_r * a.tw;
  const int ktot = a.th * a.tw * a.C;
  const int mbeg = split * a.rows_per_split;
  const int mend = (mbeg + a.rows_per_split < a.M) ? (mbeg + a.rows_per_split) : a.M;
  const int nsteps = (mend > mbeg) ? (mend - mbeg + 31) / 32 : 0;

  // dY (and X under the identity map) are bounded at THIS workgroup's last pixel row: rows past it read zeros through the
  // descriptor's range check, the tail needs no per-lane compare
  const pf_wrsrc_t rsY = PF_W_MAKE_RSRC(a.dY, (uint32_t)mend * (uint32_t)a.N * 2u);
  const pf_wrsrc_t rsX = PF_W_MAKE_RSRC(a.X, (MAPM == 0) ? (uint32_t)mend * (uint32_t)a.C * 2u : a.x_bytes);
  constexpr uint32_t OOB = 0x80000000u;
  const int sb = lane >> 4, srow = (lane & 15) >> 1, sch = lane & 1;       // role inside one LDS-DMA instruction
  const int hw_o = a.Ho * a.Wo;

  // this wavefront's instruction slots: id = wave + k * NWAVE; id < IDY: dY piece (pg, g4), else X piece (pg, g4)
  float psc[XS][8], psh[XS][8];
  Pro pro;
  pro_init(pro, a.act_lo, a.act_hi, a.kq, PRO ? a.slot : nullptr);
  if (PRO) {
#pragma unroll
    for (int x = 0; x < XS; ++x) {
      const int xi = wave + x * NWAVE;
      if (xi < IX) {
        const int g4 = xi % (TK / 64);
        const int c = c0 + (g4 * 4 + sb) * 16 + sch * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { psc[x][j] = pro_fold_scale(pro, a.ss[c + j]); psh[x][j] = pro_fold_shift(pro, a.ss[a.C + c + j]); }
      }
    }
  }

  // Source offsets of this lane's LDS-DMA slots, kept INCREMENTALLY: one add per slot and step (round 2 recomputed every
  // offset from the pixel index with two 32-bit multiplies, the strided map with two more and a divergent while loop --
  // the cycle stamps of round 3 showed ~900 of a step's 2700 cycles in that arithmetic, tools/gpu/wrw_timeline.py).
  uint32_t yoff[KDY], xoff[XS];
  const uint32_t ystep = 32u * (uint32_t)a.N * 2u, xstep = 32u * (uint32_t)a.C * 2u;
#pragma unroll
  for (int k = 0; k < KDY; ++k) {
    const int id = (wave + k * NWAVE) % IDY;
    const int pg = id / (TN / 64), g4 = id % (TN / 64);
    const int n = n0 + (g4 * 4 + sb) * 16 + sch * 8;
    yoff[k] = (n < a.N && wave + k * NWAVE < IDY) ? (uint32_t)((mbeg + pg * 8 + srow) * a.N + n) * 2u : OOB;   // (OOB + steps stays out of range)
  }
  // MAPM == 1: (ho, wo) of the lane's pixel per X slot, advanced by 32 pixels per step with ONE conditional wrap each (the
  // launcher selects this mode only when 32 / Wo + 1 <= Ho); MAPM == 2: (img, ho, wo), general
  int px_img[XS], px_ho[XS], px_wo[XS];
  const int adv_q = 32 / a.Wo, adv_r = 32 - adv_q * a.Wo;
  const int tap_dr = tap_r - a.pad_h, tap_ds = tap_s - a.pad_w;
#pragma unroll
  for (int x = 0; x < XS; ++x) {
    const int xi = (wave + x * NWAVE) % IX;
    const int pg = xi / (TK / 64), g4 = xi % (TK / 64);
    const int m = mbeg + pg * 8 + srow;
    const int cofs = c0 + (g4 * 4 + sb) * 16 + sch * 8;
    px_img[x] = 0; px_ho[x] = 0; px_wo[x] = 0;
    if (MAPM != 0) {
      const int img = m / hw_o, rem = m - img * hw_o;
      px_img[x] = img; px_ho[x] = rem / a.Wo; px_wo[x] = rem - px_ho[x] * a.Wo;
    }
    // identity: row m; shift: row m + dr * Wd + ds (may be "negative" for masked taps: the offset is only used when valid)
    const int row0 = (MAPM == 1) ? m + tap_dr * a.Wd + tap_ds : m;
    xoff[x] = (uint32_t)(row0 * a.C + cofs) * 2u;
  }
  auto stage = [&](int step, int buf) {
    unsigned char* dst = smem + buf * STAGE;
    const int mb = mbeg + step * 32;
#pragma unroll
    for (int k = 0; k < KDY; ++k) {
      const bool owny = wave + k * NWAVE < IDY;                             // wave-uniform
      const int id = (wave + k * NWAVE) % IDY;
      const int pg = id / (TN / 64), g4 = id % (TN / 64);
      unsigned char* ydst = owny ? dst + (pg * NBN + g4 * 4) * 256 : smem + 3 * STAGE;
      if (!(PF_W2_ABLATE & 2)) PF_W_BUFFER_LOAD_LDS16(rsY, ydst, yoff[k]);
      yoff[k] += ystep;
    }
#pragma unroll
    for (int x = 0; x < XS; ++x) {
      // every wavefront issues exactly XS X-pieces per stage (the counted vmcnt needs one number for all of them): a
      // wavefront without a piece of its own loads zeros (out-of-range offset) into a 1 KiB sink behind the ring
      const bool own = wave + x * NWAVE < IX;                               // wave-uniform
      const int xi = (wave + x * NWAVE) % IX;
      const int pg = xi / (TK / 64), g4 = xi % (TK / 64);
      uint32_t voff;
      if (MAPM == 0) {
        voff = xoff[x];                                                      // tail rows: descriptor range check
        xoff[x] += xstep;
      } else if (MAPM == 1) {
        const int m = mb + pg * 8 + srow;
        const bool ok = m < mend && (unsigned)(px_ho[x] + tap_dr) < (unsigned)a.H && (unsigned)(px_wo[x] + tap_ds) < (unsigned)a.Wd;
        voff = ok ? xoff[x] : OOB;
        xoff[x] += xstep;
        px_wo[x] += adv_r; px_ho[x] += adv_q;
        if (px_wo[x] >= a.Wo) { px_wo[x] -= a.Wo; ++px_ho[x]; }
        if (px_ho[x] >= a.Ho) px_ho[x] -= a.Ho;
      } else {
        const int m = mb + pg * 8 + srow;
        const int hi = px_ho[x] * a.stride + tap_r - a.pad_h, wi = px_wo[x] * a.stride + tap_s - a.pad_w;
        const bool ok = m < mend && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.Wd;
        const int row = (px_img[x] * a.H + hi) * a.Wd + wi;
        voff = ok ? (uint32_t)(row * a.C + c0 + (g4 * 4 + sb) * 16 + sch * 8) * 2u : OOB;
        px_wo[x] += adv_r; px_ho[x] += adv_q;
        if (px_wo[x] >= a.Wo) { px_wo[x] -= a.Wo; ++px_ho[x]; }
        while (px_ho[x] >= a.Ho) { px_ho[x] -= a.Ho; ++px_img[x]; }
      }
      if (!own) voff = OOB;
      unsigned char* xdst = own ? dst + DY_BYTES + (pg * NBK + g4 * 4) * 256 : smem + 3 * STAGE;
      if (!(PF_W2_ABLATE & 2)) PF_W_BUFFER_LOAD_LDS16(rsX, xdst, voff);
    }
  };
  auto transform = [&](int buf) {                                            // own X pieces, in place
    unsigned char* dst = smem + buf * STAGE;
#pragma unroll
    for (int x = 0; x < XS; ++x) {
      const int xi = wave + x * NWAVE;
      if (xi < IX) {
        const int pg = xi / (TK / 64), g4 = xi % (TK / 64);
#pragma unroll
        for (int j = 0; j < 8; ++j) { pro.sc[j] = psc[x][j]; pro.sh[j] = psh[x][j]; }
        const uint32_t p = lds_addr(dst + DY_BYTES + (pg * NBK + g4 * 4) * 256 + lane * 16);   // asm accesses: see pf_conv_common.h
        lds_write_b128(p, pro_apply(pro, lds_read_b128(p)));
      }
    }
  };

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int tr_off = (q & 1) * 128 + (l15 >> 2) * 32 + (l15 & 3) * 8;
  const int pg_lo = q >> 1;

  // Three-stage ring with COUNTED waits: the loads of step t+2 are issued at the top of step t and have two steps to land;
  // a wavefront waits for its own pieces of step t+1 (vmcnt(LPS): the youngest stage stays in flight across the barrier),
  // transforms them in place (PRO), and the barrier publishes them.  One barrier per step.
  constexpr int NSTG = 3, LPS = KDY + XS;
  if (nsteps > 0) stage(0, 0);
  if (nsteps > 1) { stage(1, 1); wrw_wait_vm<LPS>(); } else wrw_wait_vm<0>();
  if (PRO && nsteps > 0) transform(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0, ibuf = 2;
#ifdef PF_W2_TIMING
  const int tm_blk = gridDim.x / 2;
  uint32_t* tm_out = reinterpret_cast<uint32_t*>(a.slabs) + wave * 64;
#endif
  for (int t = 0; t < nsteps; ++t) {
    const bool more2 = t + 2 < nsteps;
    PF_W2_STAMP(0);
    if (more2) { stage(t + 2, ibuf); ibuf = (ibuf + 1 == NSTG) ? 0 : ibuf + 1; }
    PF_W2_STAMP(1);
    const unsigned char* sbase = smem + buf * STAGE;
    buf = (buf + 1 == NSTG) ? 0 : buf + 1;                                  // now: the buffer of step t+1
    if (!(PF_W2_ABLATE & 1)) {
      // fragments by transposing LDS reads issued from inline asm (tr_read_frags): the compiler must not see them, or it
      // drains the LDS-DMA of the younger stages (vmcnt(0)) in front of the first one
      bf16x8 xf[NJ], df[NI];
      if constexpr (NI == 4 && NJ == 4) {
        tr_read_frags2<2 * NBK * 256, 2 * NBN * 256>(lds_addr(sbase + DY_BYTES + (pg_lo * NBK + wk * NJ) * 256 + tr_off),
                                                     lds_addr(sbase + (pg_lo * NBN + wn * NI) * 256 + tr_off), xf, df);
      } else
      {
      tr_read_frags<2 * NBK * 256>(lds_addr(sbase + DY_BYTES + (pg_lo * NBK + wk * NJ) * 256 + tr_off), xf);
      tr_read_frags<2 * NBN * 256>(lds_addr(sbase + (pg_lo * NBN + wn * NI) * 256 + tr_off), df);
      }
      PF_W2_STAMP(2);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[i], xf[j], acc[i][j], 0, 0, 0);
    }
    // the MFMAs are register-only: without this fence the compiler sinks them BELOW the wait (inline asm's "memory" clobber
    // orders memory operations only), and the wavefront would sit out the load latency before starting its matrix work
    __builtin_amdgcn_sched_barrier(0);
    PF_W2_STAMP(3);
    if (t + 1 < nsteps) {
      if (more2) wrw_wait_vm<LPS>(); else wrw_wait_vm<0>();                 // own pieces of step t+1 have landed
      PF_W2_STAMP(4);
      if (PRO) transform(buf);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PF_W2_STAMP(5);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PF_W2_STAMP(6);
  }
#ifdef PF_W2_TIMING
  return;
#endif

  float* out = a.slabs + (int64_t)split * a.N * ktot;
  const int kcol = tap * a.C + c0 + wk * WTK;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = kcol + j * 16 + l15;
      const int n = n0 + wn * WTN + i * 16 + q * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.N && (!(PF_W2_ABLATE & 4) || acc[i][j][r] == 12345.678f)) out[(int64_t)(n + r) * ktot + k] = acc[i][j][r];
    }
}

struct Wrw2Cfg { int tn, tk; };
static Wrw2Cfg wrw2_pick(int N, int C) {
  const int tn = (N % 256 == 0) ? 256 : ((N % 128 == 0) ? 128 : 64);
  int tk = (C % 128 == 0) ? 128 : 64;
  // round 6: few output channels over many input channels (conv1 of stages 1-2: 256 -> 64, 256 / 512 -> 128): 256 input channels per
  // tile -- dY is staged once per 256 channels instead of once per 128, and a workgroup's pixel range is half as long at the same
  // workgroup count (these launches are bound by their per-step cost: 166 us against an 82 us HBM floor, profiles/r06_wrw_layers.txt)
  if (tn <= 128 && C % 256 == 0 && pf_tuning().wrw2_tk256 != 0) tk = 256;
  if (tn == 128 && tk == 64) return Wrw2Cfg{64, 64};     // (128, 64) has no instantiation: two 64-wide tiles
  return Wrw2Cfg{tn, tk};
}

static bool wrw2_enabled() {
  return pf_tuning().wrw2 != 0;                            // PF_WRW2=0: tuning / A-B override
}

// pixel splits of the shared-tile kernel (0: does not apply)
int pf_wrw2_splits(int M, int N, int C, int taps) {
  if (!wrw2_enabled() || (C % 64) || (N % 64) || M < 2048) return 0;
  if ((int64_t)M * N >= ((int64_t)1 << 30)) return 0;
  const Wrw2Cfg c = wrw2_pick(N, C);
  const int tiles = (N / c.tn) * (taps * C / c.tk);
  // workgroups a launch aims at: every pixel split writes (and the reduction reads) one fp32 slab of the whole dW tile, so
  // more splits buy parallelism with HBM traffic.  Measured per layer over 128 ... 1024 (profiles/r03_wrw_target_bench.txt):
  // one workgroup per CU for the 1x1 layers, 1.5 for the 3x3 ones (round 2's two per CU cost 0.35 ms per ResNet-50 step).
  const int forced = pf_tuning().wrw2_target;             // PF_WRW2_TARGET: tuning / A-B override
  const int target = (forced > 0) ? forced : (taps == 1 ? 256 : 384);
  int S = (target + tiles - 1) / tiles;
  const int maxS = (M + 255) / 256;                       // >= 8 steps of 32 pixels per workgroup
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  int rows = (M + S - 1) / S;
  rows = ((rows + 31) / 32) * 32;
  return (M + rows - 1) / rows;
}

template <int TN, int TK, int WTN, int WTK, bool PRO, int MAPM>
static int wrw2_launch_t(const Wrw2Args& a, int grid, hipStream_t st) {
  const size_t lds = 3 * (size_t)(4 * (TN / 16) * 256 + 4 * (TK / 16) * 256) + 1024;   // three stages + the 1 KiB sink
  if (int e = pf_require_lds(reinterpret_cast<const void*>(&k_wrw2<TN, TK, WTN, WTK, PRO, MAPM>), lds)) return e;
  k_wrw2<TN, TK, WTN, WTK, PRO, MAPM><<<grid, 64 * (TN / WTN) * (TK / WTK), lds, st>>>(a);
  PF_LAUNCH_CHECK();
  return 0;
}

int pf_wrw2_launch(const void* dY, const void* X, float* slabs, const float* scale_shift, int act, const uint32_t* slot,
                   int bits, int M, int N, int C, int th, int tw, int H, int Wd, int Ho, int Wo, int stride, int pad_h,
                   int pad_w, int S, int64_t x_rows, hipStream_t st) {
  if (S <= 0) return -1;
  if (x_rows * C >= ((int64_t)1 << 30) || (int64_t)M * N >= ((int64_t)1 << 30)) return -1;   // 31-bit byte offsets inside the kernel
  Wrw2Args a;
  a.dY = (const bf16_t*)dY; a.X = (const bf16_t*)X; a.slabs = slabs; a.ss = scale_shift; a.slot = slot;
  a.kq = uq_k_of_bits(slot ? bits : 8);
  a.act_lo = (act == PF_ACT_NONE) ? -INFINITY : 0.0f;
  a.act_hi = (act == PF_ACT_RELU6) ? 6.0f : INFINITY;
  a.dy_bytes = (uint32_t)((int64_t)M * N * 2);
  a.x_bytes = (uint32_t)(x_rows * C * 2);
  a.M = M; a.N = N; a.C = C; a.th = th; a.tw = tw; a.H = H; a.Wd = Wd; a.Ho = Ho; a.Wo = Wo;
  a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
  const Wrw2Cfg c = wrw2_pick(N, C);
  a.tiles_n = N / c.tn;
  a.tiles = a.tiles_n * (th * tw * C / c.tk);
  int rows = (M + S - 1) / S;
  rows = ((rows + 31) / 32) * 32;
  a.rows_per_split = rows;
  const int grid = a.tiles * S;
  const bool pro = scale_shift != nullptr;
  // 0: identity; 1: stride-1 tap of a same-size convolution (incremental shift addressing; one wrap per step must be
  // enough: 32 / Wo + 1 <= Ho); 2: general
  int mapm = 0;
  if (stride != 1 || th * tw > 1) mapm = (stride == 1 && Ho == H && Wo == Wd && 32 / Wo + 1 <= Ho) ? 1 : 2;
  if (pro && th * tw > 1) return -1;
  if (pro && mapm == 1) mapm = 2;                       // (prologue variants: identity and general only)
#define PF_W2(TNV, TKV, WTNV, WTKV)                                                                           \
  do {                                                                                                        \
    if (pro) return mapm ? wrw2_launch_t<TNV, TKV, WTNV, WTKV, true, 2>(a, grid, st)                           \
                         : wrw2_launch_t<TNV, TKV, WTNV, WTKV, true, 0>(a, grid, st);                          \
    if (mapm == 1) return wrw2_launch_t<TNV, TKV, WTNV, WTKV, false, 1>(a, grid, st);                          \
    return mapm ? wrw2_launch_t<TNV, TKV, WTNV, WTKV, false, 2>(a, grid, st)                                   \
                : wrw2_launch_t<TNV, TKV, WTNV, WTKV, false, 0>(a, grid, st);                                  \
  } while (0)
  if (c.tn == 128 && c.tk == 256) PF_W2(128, 256, 64, 64);
  if (c.tn == 64 && c.tk == 256) PF_W2(64, 256, 32, 64);
  if (c.tn == 256 && c.tk == 128) PF_W2(256, 128, 64, 64);
  if (c.tn == 128 && c.tk == 128) PF_W2(128, 128, 64, 64);
  if (c.tn == 256 && c.tk == 64) PF_W2(256, 64, 64, 32);
  if (c.tn == 64 && c.tk == 128) PF_W2(64, 128, 32, 64);
  PF_W2(64, 64, 32, 32);
#undef PF_W2
}

// ---- RxS backward-filter behind the C ABI ------------------------------------------------------------------------
int pf_wrw_reduce(float* workspace, int S, int64_t n, void* dW, int dw_dtype, hipStream_t st);   // pf_conv.hip

// pixel splits of pf_conv2d_wrw (0: shape not supported); the workspace must hold (splits + 32) * N * th*tw*C floats
// pf_wrw3x3_c64.hip: the window-staged kernel for 3x3 / stride 1, 64 -> 64 channels on 56 x 56 maps
bool pf_wrw3x3_c64_geom(int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w, int Ho, int Wo);
int pf_wrw3x3_c64_splits(int imgs);
int pf_wrw3x3_c64_launch(const void* dY, const void* X, float* slabs, int imgs, hipStream_t st);

extern "C" int pf_conv2d_wrw_splits(int M, int N, int C, int taps) {
  const int s2 = pf_wrw2_splits(M, N, C, taps);
  int s = s2 > 0 ? s2 : pf_wrw_tr_splits(M, N, C, taps);
  // this query does not see the image size: where the window-staged kernel COULD take the launch (its geometry is 56 x 56 images),
  // the answer covers its slab count too -- an upper bound for sizing the workspace; pf_conv2d_wrw folds what it actually wrote
  if (pf_tuning().conv3x3_c64 != 0 && taps == 9 && N == 64 && C == 64 && M % (56 * 56) == 0) {
    const int s3 = pf_wrw3x3_c64_splits(M / (56 * 56));
    if (s3 > s) s = s3;
  }
  return s;
}

// dW[n][r][s][c] = sum_m dY[m][n] * X[pix(m, r, s)][c]  (KRSC, float32 or bf16), X a materialised NHWC activation
extern "C" int pf_conv2d_wrw(const void* dY, const void* X, void* dW, int dw_dtype, float* workspace, int imgs, int H,
                             int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w, int Ho, int Wo,
                             void* stream) {
  if (imgs <= 0 || H <= 0 || Wd <= 0 || Ho <= 0 || Wo <= 0 || th < 1 || tw < 1 || stride < 1) return (int)hipErrorInvalidValue;
  if (!pf_aligned16(dY) || !pf_aligned16(X) || !pf_aligned16(dW) || !pf_aligned16(workspace)) return (int)hipErrorInvalidValue;
  const int M = imgs * Ho * Wo;
  hipStream_t st = (hipStream_t)stream;
  if (pf_wrw3x3_c64_geom(H, Wd, C, N, th, tw, stride, pad_h, pad_w, Ho, Wo)) {
    const int r3 = pf_wrw3x3_c64_launch(dY, X, workspace, imgs, st);
    if (r3 == 0) return pf_wrw_reduce(workspace, pf_wrw3x3_c64_splits(imgs), (int64_t)N * th * tw * C, dW, dw_dtype, st);
    if (r3 > 0) return r3;
  }
  const int s2 = pf_wrw2_splits(M, N, C, th * tw);
  const int S = s2 > 0 ? s2 : pf_wrw_tr_splits(M, N, C, th * tw);
  if (S <= 0) return (int)hipErrorInvalidValue;
  int r = -1;
  if (s2 > 0)
    r = pf_wrw2_launch(dY, X, workspace, nullptr, PF_ACT_NONE, nullptr, 8, M, N, C, th, tw, H, Wd, Ho, Wo, stride, pad_h,
                       pad_w, S, (int64_t)imgs * H * Wd, st);
  if (r < 0)
    r = pf_wrw_tr_launch(dY, X, workspace, nullptr, PF_ACT_NONE, nullptr, 8, M, N, C, th, tw, H, Wd, Ho, Wo, stride,
                         pad_h, pad_w, S, st);
  if (r != 0) return r < 0 ? (int)hipErrorInvalidValue : r;
  return pf_wrw_reduce(workspace, S, (int64_t)N * th * tw * C, dW, dw_dtype, st);
}
