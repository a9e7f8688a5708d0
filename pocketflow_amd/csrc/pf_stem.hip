// The ResNet stem: 7x7 / stride 2 / pad 3 convolution of a 3-channel image to 64 channels
// (reference utils/external/resnet_model.py:478-486 initial conv2d_fixed_padding), NHWC bf16 -> NHWC bf16.
//
// Why its own kernel.  With C = 3 the implicit-GEMM kernel's "k = (tap, 64 channels)" tiling does not apply, and MIOpen
// spends 384 us on the 256 x 224 x 224 launch (77 MB in, 411 MB out: HBM floor ~80 us).  Layout of the problem on CDNA4:
//   * a workgroup owns one image and a strip of 8 output rows; the 21 input rows it needs are staged ONCE into LDS with
//     every pixel padded from 3 to 4 channels (8 bytes), 3 zero pixels left of the row and >= 5 right of it.  Then the
//     eight k-values a lane feeds to v_mfma_f32_16x16x32_bf16 -- taps (s, s+1) x 4 channels of kernel row r -- are ONE
//     aligned 16-byte LDS read at  row(2*oy + r) + (2*ox + s) * 8 B:  k' = r*32 + s*4 + c, 7 k-steps of 32 with the
//     weights of s = 7 and of c = 3 being zero (147 real of 224 k-values: 34 % padding, bought for conflict-free,
//     division-free, shuffle-free operand fetch);
//   * weights are operand A (rows = output channels), packed once per workgroup into the same k' order and kept in
//     REGISTERS for the whole launch (4 channel blocks x 7 k-steps x 4 VGPRs); the MFMA row -> channel map is chosen
//     such that a lane ends up with 16 CONSECUTIVE channels of its pixel: two 16-byte stores, no LDS transposition;
//   * persistent workgroups (2 per CU) loop over (image, strip) items, so the weight packing is paid once.
#include "pf_conv_common.h"

#define ST_THREADS 256
#define ST_OROWS 8                      // output rows per item
#define ST_IROWS (2 * ST_OROWS + 5)     // input rows staged per item
#define ST_N 64
#define ST_R 7
#define ST_WL_BYTES (ST_N * ST_R * 32 * 2)

struct StemArgs {
  const bf16_t* X;   // [imgs][H][Wd][3]
  const bf16_t* W;   // [64][7][7][3]
  bf16_t* Y;         // [imgs][Ho][Wo][64]
  int imgs, H, Wd, Ho, Wo;
  int strips, n_items;
  int rsb;           // LDS bytes per staged input row = (Wd + 8) * 8
};

__global__ __launch_bounds__(ST_THREADS) void k_stem7x7_fwd(StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;                               // packed weights [64][7][32] bf16
  unsigned char* xs = smem + ST_WL_BYTES;                 // staged rows [21][(Wd + 8)][4] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;

  // ---- pack the weights: chunk (n, r, q) = taps (2q, 2q+1) x 4 channels of kernel row r of output channel n ----
  for (int ch = tid; ch < ST_N * ST_R * 4; ch += ST_THREADS) {
    const int q = ch & 3, nr = ch >> 2;                   // nr = n*7 + r
    const bf16_t* src = a.W + ((int64_t)nr * 7 + 2 * q) * 3;
    uint32_t v[4];
    const uint32_t e0 = src[0], e1 = src[1], e2 = src[2];
    v[0] = e0 | (e1 << 16); v[1] = e2;
    if (q < 3) {
      const uint32_t f0 = src[3], f1 = src[4], f2 = src[5];
      v[2] = f0 | (f1 << 16); v[3] = f2;
    } else {
      v[2] = 0u; v[3] = 0u;                               // tap s = 7 does not exist
    }
    *reinterpret_cast<uint4*>(wl + ((int64_t)nr * 32 + q * 8) * 2) = make_uint4(v[0], v[1], v[2], v[3]);
  }
  // ---- zero the halo pixels of every staged row once (columns 0..2 and Wd+3..Wd+7 are never written again) ----
  for (int t = tid; t < ST_IROWS * 8; t += ST_THREADS) {
    const int row = t >> 3, h = t & 7;
    const int pp = h < 3 ? h : a.Wd + h;                  // 0,1,2, Wd+3 .. Wd+7
    *reinterpret_cast<uint2*>(xs + (int64_t)row * a.rsb + pp * 8) = make_uint2(0u, 0u);
  }
  __syncthreads();
  // ---- A fragments: MFMA row i of channel block nb is output channel 16*(i/4) + 4*nb + (i%4) ----
  bf16x8 wf[4][ST_R];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = 16 * (l16 >> 2) + 4 * nb + (l16 & 3);
#pragma unroll
    for (int r = 0; r < ST_R; ++r)
      wf[nb][r] = *reinterpret_cast<const bf16x8*>(wl + ((n * ST_R + r) * 32 + g * 8) * 2);
  }

  const int pairs = a.Wd >> 1;                            // two pixels (12 bytes) per staging task
  const int pblocks = a.Wo >> 4;                          // 16-pixel blocks per output row
  const int units = ST_OROWS * pblocks;
  for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
    const int img = item / a.strips, strip = item - img * a.strips;
    const int oy0 = strip * ST_OROWS, iy0 = 2 * oy0 - 3;
    __syncthreads();                                      // the previous item's fragment reads are done
    // ---- stage 21 input rows, 3 -> 4 channels ----
    for (int row = tid >> 7; row < ST_IROWS; row += 2) {
      const int iy = iy0 + row;
      unsigned char* dst = xs + (int64_t)row * a.rsb + 3 * 8;
      const bool live = iy >= 0 && iy < a.H;
      const uint32_t* src = reinterpret_cast<const uint32_t*>(a.X + ((int64_t)img * a.H + (live ? iy : 0)) * a.Wd * 3);
      for (int j = tid & 127; j < pairs; j += 128) {
        uint32_t d0 = 0u, d1 = 0u, d2 = 0u;
        if (live) { d0 = src[3 * j]; d1 = src[3 * j + 1]; d2 = src[3 * j + 2]; }
        // (the first real pixel sits 24 bytes into the row: 8-byte stores)
        *reinterpret_cast<uint2*>(dst + j * 16) = make_uint2(d0, d1 & 0xFFFFu);
        *reinterpret_cast<uint2*>(dst + j * 16 + 8) = make_uint2((d1 >> 16) | (d2 << 16), d2 >> 16);
      }
    }
    __syncthreads();
    // ---- 16-pixel x 64-channel units ----
    for (int u = wave; u < units; u += ST_THREADS / 64) {
      const int orow = u / pblocks, pb = u - orow * pblocks;
      const int oy = oy0 + orow;
      if (oy >= a.Ho) continue;
      const int ox = pb * 16 + l16;
      const unsigned char* base = xs + (int64_t)(2 * orow) * a.rsb + (2 * ox + 2 * g) * 8;
      f32x4 acc[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < ST_R; ++r) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(base + (int64_t)r * a.rsb);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][r], xf, acc[nb], 0, 0, 0);
      }
      // lane: pixel ox, channels 16g + 4nb + j
      bf16_t* out = a.Y + (((int64_t)img * a.Ho + oy) * a.Wo + ox) * ST_N + 16 * g;
      const uint4 lo = make_uint4(pack_bf16x2(acc[0][0], acc[0][1]), pack_bf16x2(acc[0][2], acc[0][3]),
                                  pack_bf16x2(acc[1][0], acc[1][1]), pack_bf16x2(acc[1][2], acc[1][3]));
      const uint4 hi = make_uint4(pack_bf16x2(acc[2][0], acc[2][1]), pack_bf16x2(acc[2][2], acc[2][3]),
                                  pack_bf16x2(acc[3][0], acc[3][1]), pack_bf16x2(acc[3][2], acc[3][3]));
      *reinterpret_cast<uint4*>(out) = lo;
      *reinterpret_cast<uint4*>(out + 8) = hi;
    }
  }
}

// 1 when pf_conv_stem_fwd takes the shape (anything else stays with the caller's generic path)
extern "C" int pf_conv_stem_supported(int H, int Wd, int C, int N, int k, int stride, int pad) {
  return (C == 3 && N == ST_N && k == 7 && stride == 2 && pad == 3 && H > 0 && (H % 2) == 0 && Wd >= 32 && (Wd % 32) == 0 &&
          Wd <= 1024) ? 1 : 0;
}

extern "C" int pf_conv_stem_fwd(const void* X, const void* W, void* Y, int imgs, int H, int Wd, void* stream) {
  if (!pf_conv_stem_supported(H, Wd, 3, ST_N, 7, 2, 3) || imgs <= 0) return (int)hipErrorInvalidValue;
  if (!pf_aligned16(Y) || (reinterpret_cast<uintptr_t>(X) & 3u) || (reinterpret_cast<uintptr_t>(W) & 1u))
    return (int)hipErrorInvalidValue;
  StemArgs a;
  a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.Y = (bf16_t*)Y;
  a.imgs = imgs; a.H = H; a.Wd = Wd; a.Ho = H / 2; a.Wo = Wd / 2;
  a.strips = (a.Ho + ST_OROWS - 1) / ST_OROWS;
  a.n_items = imgs * a.strips;
  a.rsb = (Wd + 8) * 8;
  const size_t lds = (size_t)ST_WL_BYTES + (size_t)ST_IROWS * a.rsb;
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  static size_t configured = 0;
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem7x7_fwd),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    configured = lds;
  }
  const int grid = a.n_items < 512 ? a.n_items : 512;
  k_stem7x7_fwd<<<grid, ST_THREADS, lds, (hipStream_t)stream>>>(a);
  PF_LAUNCH_CHECK();
  return 0;
}
