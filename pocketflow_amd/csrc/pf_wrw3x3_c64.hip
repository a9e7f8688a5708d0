// K12, backward-filter of the geometry pf_conv3x3_c64.hip serves forward: 3x3 / stride 1 / pad 1, 64 -> 64 channels on 56 x 56 maps
// (conv2 of ResNet-50's first stage; Conv2DBackpropFilter of utils/external/resnet_model.py:92-103).
//   dW[n][r][s][c] = sum over the 802 816 output pixels p of  dY[p][n] * X[p + (r - 1, s - 1)][c]
//
// The shared-tile kernel of pf_wrw.hip (k_wrw2<64, 64, 32, 32>) gives this layer one workgroup per (tap, pixel split): every tap
// re-stages BOTH operands, a wavefront runs 4 MFMAs per 8 KiB step -- 211-233 us against a 24 us MFMA / 33 us HBM floor and MIOpen's
// 165 us (profiles/r06_wrw_layers_before.txt), the worst row of the backward-filter table.  Here a workgroup walks tiles of TWO image
// rows (112 pixels) and computes all nine taps of a tile from ONE staging of its operands:
//   * LDS: the input window of pf_conv3x3_c64.hip (4 rows x 58 columns x 64 channels, pixels 144 bytes apart, padding rows / columns
//     zero) + the dY tile (112 pixels x 64 channels, same pitch, rows 112-127 zero): 51 KiB, two workgroups per CU;
//   * the contraction runs over PIXELS, the slow index of both tensors: fragments come from transposing LDS reads
//     (ds_read_b64_tr_b16, as in pf_wrw.hip) -- each lane supplies the address of ITS pixel row, so the pixel pitch, the row wrap of
//     the window and the tap shift ((r * 58 + s) * 144 bytes, an immediate offset) cost nothing;
//   * the [64] x [9 x 64] output of the workgroup stays in registers across all of its tiles: 4 x 36 accumulator blocks, one 16-channel block of the input x
//     nine taps per wavefront = 144 registers; per 32-pixel k-block a wavefront reads 4 dY and 9 X fragments
//     for 36 MFMAs;
//   * one float32 slab [64][576] per workgroup at the end, folded by pf_wrw_reduce in a fixed order (deterministic).
#include "pf_conv_common.h"
#include "pf_igemm.h"

typedef short w3_v4s __attribute__((ext_vector_type(4)));
typedef short w3_v8s __attribute__((ext_vector_type(8)));

#define W3_W 56
#define W3_C 64
#define W3_TP 112
#define W3_WCOLS 58
#define W3_PITCH 144
#define W3_WPIX (4 * W3_WCOLS)
#define W3_NX 33                                 // LDS-DMA instructions of the input window (232 x 144 B -> 33 KiB)
#define W3_NDY 18                                // ... of the dY tile (128 rows x 144 B = 18 KiB exactly)
#define W3_XBYTES (W3_NX * 1024)
#define W3_LDS (W3_XBYTES + W3_NDY * 1024)
#define W3_THREADS 256
#define W3_KTOT (9 * W3_C)

struct Wrw3Args {
  const bf16_t* dY;     // [M][64]
  const bf16_t* X;      // [M][64] (NHWC, 56 x 56 images)
  float* slabs;         // [gridDim.x][64][576]
  uint32_t bytes;       // size of dY = size of X in bytes
  int n_tiles;
};

template <int OFF>
__device__ __forceinline__ void w3_tr(uint32_t p, w3_v4s& v) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(v) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ bf16x8 w3_join(const w3_v4s& lo, const w3_v4s& hi) {
  w3_v8s x;
  x[0] = lo[0]; x[1] = lo[1]; x[2] = lo[2]; x[3] = lo[3];
  x[4] = hi[0]; x[5] = hi[1]; x[6] = hi[2]; x[7] = hi[3];
  return *reinterpret_cast<const bf16x8*>(&x);
}
__device__ __forceinline__ void w3_wait8(w3_v4s (&v)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
}
__device__ __forceinline__ void w3_wait6(w3_v4s (&v)[6]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) : : "memory");
}

// three taps E .. E + 2 of a wavefront's 16 input channels in k-block KB: 6 transposing reads of X, 12 MFMAs.  (A wavefront owns ONE
// 16-channel block of the input for all nine taps: the tap shift is an immediate offset common to all wavefronts, the channel block
// sits in the base address -- with per-wavefront immediates the four instantiations of the tile body cost 565 spilled registers.)
template <int KB, int E>
__device__ __forceinline__ void w3_cols(const uint32_t (&xb)[4][2], const bf16x8 (&df)[4], f32x4 (&acc)[4][9]) {
  w3_v4s v[6];
#define W3_OFF(e) ((((e) / 3) * W3_WCOLS + (e) % 3) * W3_PITCH)
  w3_tr<W3_OFF(E)>(xb[KB][0], v[0]);     w3_tr<W3_OFF(E)>(xb[KB][1], v[1]);
  w3_tr<W3_OFF(E + 1)>(xb[KB][0], v[2]); w3_tr<W3_OFF(E + 1)>(xb[KB][1], v[3]);
  w3_tr<W3_OFF(E + 2)>(xb[KB][0], v[4]); w3_tr<W3_OFF(E + 2)>(xb[KB][1], v[5]);
#undef W3_OFF
  w3_wait6(v);
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const bf16x8 xf = w3_join(v[2 * e], v[2 * e + 1]);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][E + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[i], xf, acc[i][E + e], 0, 0, 0);
  }
}

template <int KB>
__device__ __forceinline__ void w3_kblock(uint32_t dyb, const uint32_t (&xb)[4][2], f32x4 (&acc)[4][9]) {
  // dY fragments of the four 16-channel blocks: pixel rows kb * 32 + {q * 4 + (l15 >> 2)} (lo) and + 16 (hi)
  w3_v4s d[8];
  w3_tr<KB * 32 * W3_PITCH + 0>(dyb, d[0]);   w3_tr<KB * 32 * W3_PITCH + 16 * W3_PITCH + 0>(dyb, d[1]);
  w3_tr<KB * 32 * W3_PITCH + 32>(dyb, d[2]);  w3_tr<KB * 32 * W3_PITCH + 16 * W3_PITCH + 32>(dyb, d[3]);
  w3_tr<KB * 32 * W3_PITCH + 64>(dyb, d[4]);  w3_tr<KB * 32 * W3_PITCH + 16 * W3_PITCH + 64>(dyb, d[5]);
  w3_tr<KB * 32 * W3_PITCH + 96>(dyb, d[6]);  w3_tr<KB * 32 * W3_PITCH + 16 * W3_PITCH + 96>(dyb, d[7]);
  w3_wait8(d);
  bf16x8 df[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) df[i] = w3_join(d[2 * i], d[2 * i + 1]);
  w3_cols<KB, 0>(xb, df, acc);
  w3_cols<KB, 3>(xb, df, acc);
  w3_cols<KB, 6>(xb, df, acc);
}

__device__ __forceinline__ void w3_tile(uint32_t dyb, const uint32_t (&xb)[4][2], f32x4 (&acc)[4][9]) {
  w3_kblock<0>(dyb, xb, acc);
  w3_kblock<1>(dyb, xb, acc);
  w3_kblock<2>(dyb, xb, acc);
  w3_kblock<3>(dyb, xb, acc);
}

__global__ __launch_bounds__(W3_THREADS, 2) void k_wrw3x3_c64(const Wrw3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // input window | dY tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const pf_rsrc_t rsX = PF_MAKE_RSRC(a.X, a.bytes);
  const pf_rsrc_t rsY = PF_MAKE_RSRC(a.dY, a.bytes);
  constexpr uint32_t OOB = 0x80000000u;

  // LDS-DMA: instruction I = wave + 4 * d; I < 33: input window (as pf_conv3x3_c64.hip), 33 <= I < 51: dY rows (rows >= 112: zeros)
  auto stage = [&](int t) {
    const int img = t / (W3_W / 2), h0 = (t - img * (W3_W / 2)) * 2;
    const int base = (img * (W3_W * W3_W) + h0 * W3_W) * (W3_C * 2);
    const bool top_ok = h0 > 0, bot_ok = h0 + 2 < W3_W;
#pragma unroll
    for (int d = 0; d < 13; ++d) {
      const int I = wave + 4 * d;                                            // wave-uniform
      if (I < W3_NX) {
        const int byte = I * 1024 + lane * 16;
        const int wp = byte / W3_PITCH, ck = (byte - wp * W3_PITCH) >> 4;
        const int wr = wp / W3_WCOLS, wc = wp - wr * W3_WCOLS;
        const bool ok = wp < W3_WPIX && ck < 8 && wc > 0 && wc < W3_WCOLS - 1 && (wr > 0 || top_ok) && (wr < 3 || bot_ok);
        const uint32_t voff = ok ? (uint32_t)(base + ((wr - 1) * W3_W + (wc - 1)) * (W3_C * 2) + ck * 16) : OOB;
        PF_BUFFER_LOAD_LDS16(rsX, smem + I * 1024, voff, 0);
      } else if (I < W3_NX + W3_NDY) {
        const int byte = (I - W3_NX) * 1024 + lane * 16;
        const int row = byte / W3_PITCH, ck = (byte - row * W3_PITCH) >> 4;
        const bool ok = row < W3_TP && ck < 8;
        const uint32_t voff = ok ? (uint32_t)(base + row * (W3_C * 2) + ck * 16) : OOB;
        PF_BUFFER_LOAD_LDS16(rsY, smem + I * 1024, voff, 0);
      }
    }
  };

  // fragment bases.  A transposing read takes, from lane (q, l15), 8 bytes of pixel row q * 4 + (l15 >> 2) (+ 16 for the second read of
  // a fragment) at channel quad (l15 & 3) and leaves every lane with four pixels of ONE channel: dY and X use the same pixel order, so
  // the contraction is consistent.  dY rows are linear in the pixel; X rows go through the window map (row wrap at pixel 56), and
  // pixels past the tile (the zero rows of dY) are clamped to the last window pixel: 0 x finite.
  const int pr = q * 4 + (l15 >> 2);
  const uint32_t dyb = lds_addr(smem) + (uint32_t)(W3_XBYTES + pr * W3_PITCH + (l15 & 3) * 8);
  uint32_t xb[4][2];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
      int p = kb * 32 + hi * 16 + pr;
      p = (p < W3_TP) ? p : (W3_TP - 1);
      const int wp0 = (p / W3_W) * W3_WCOLS + (p % W3_W);
      xb[kb][hi] = lds_addr(smem) + (uint32_t)(wp0 * W3_PITCH + wave * 32 + (l15 & 3) * 8);   // + this wavefront's 16-channel block
    }

  f32x4 acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 9; ++e) acc[i][e] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    stage(t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    w3_tile(dyb, xb, acc);
    __builtin_amdgcn_s_barrier();                                            // every wavefront is done with the tile: it may be overwritten
    __builtin_amdgcn_sched_barrier(0);
  }

  // slab [64][576]: accumulator block (i, e) holds rows n = i * 16 + q * 4 + r, column k = tap e * 64 + channel wave * 16 + l15
  float* out = a.slabs + (int64_t)blockIdx.x * W3_C * W3_KTOT;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      const int k = e * W3_C + wave * 16 + l15;
      const int n = i * 16 + q * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(int64_t)(n + r) * W3_KTOT + k] = acc[i][e][r];
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
bool pf_wrw3x3_c64_geom(int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w, int Ho, int Wo) {
  return pf_tuning().conv3x3_c64 != 0 && th == 3 && tw == 3 && stride == 1 && pad_h == 1 && pad_w == 1 && C == W3_C && N == W3_C &&
         H == W3_W && Wd == W3_W && Ho == W3_W && Wo == W3_W;
}

// slabs of a launch over `imgs` images (= its workgroups: two per CU, never more than tiles)
int pf_wrw3x3_c64_splits(int imgs) {
  const int n_tiles = imgs * (W3_W / 2);
  return n_tiles < 512 ? n_tiles : 512;
}

int pf_wrw3x3_c64_launch(const void* dY, const void* X, float* slabs, int imgs, hipStream_t st) {
  if ((int64_t)imgs * W3_W * W3_W * W3_C * 2 >= ((int64_t)1 << 31)) return -1;
  Wrw3Args a;
  a.dY = (const bf16_t*)dY; a.X = (const bf16_t*)X; a.slabs = slabs;
  a.bytes = (uint32_t)((int64_t)imgs * W3_W * W3_W * W3_C * 2);
  a.n_tiles = imgs * (W3_W / 2);
  if (int e = pf_require_lds(reinterpret_cast<const void*>(&k_wrw3x3_c64), W3_LDS)) return e;
  k_wrw3x3_c64<<<pf_wrw3x3_c64_splits(imgs), W3_THREADS, W3_LDS, st>>>(a);
  PF_LAUNCH_CHECK();
  return 0;
}
