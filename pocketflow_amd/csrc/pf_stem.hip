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

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef short st_v4s __attribute__((ext_vector_type(4)));
typedef short st_v8s __attribute__((ext_vector_type(8)));
// device-only buffer-descriptor type: the host pass only has to parse the kernel bodies (see pf_igemm.hip)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t st_rsrc_t;
#define ST_MAKE_RSRC(p, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(p), (short)0, (int)(bytes), 0x00020000)
#define ST_BUFFER_LOAD_LDS16(rs, lds, voff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, voff, 0, 0, 0)
#else
typedef int st_rsrc_t;
#define ST_MAKE_RSRC(p, bytes) 0
#define ST_BUFFER_LOAD_LDS16(rs, lds, voff) ((void)(rs), (void)(lds), (void)(voff))
#endif

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
  if (int e = pf_require_lds(reinterpret_cast<const void*>(&k_stem7x7_fwd), lds)) return e;
  const int grid = a.n_items < 512 ? a.n_items : 512;
  k_stem7x7_fwd<<<grid, ST_THREADS, lds, (hipStream_t)stream>>>(a);
  PF_LAUNCH_CHECK();
  return 0;
}


// =================================================================================================================
// Backward-filter of the stem:  dW[n][r][s][c] = sum over (img, oy, ox) of dY[img][oy][ox][n] * X[img][2oy+r-3][2ox+s-3][c]
// (Conv2DBackpropFilter of the same convolution).  Both operands are pixel-major in memory and the contraction runs over
// pixels, so both MFMA operands are fetched with the transposing LDS read ds_read_b64_tr_b16 (a lane supplies the address
// of a 4-element piece and receives a column):
//   * dY^T: two output rows of one image (2*Wo consecutive pixels) go to LDS by LDS-DMA in [8 pixels][16 channels]
//     blocks of 256 B, exactly the layout of k_wrw2 (pf_wrw.hip);
//   * X: the 9 input rows of the strip are staged as in the forward kernel (pixels padded to 4 channels).  A piece of
//     the transposed read -- 4 consecutive k' = (tap s, channels 0..3) of one pixel -- is then ONE padded pixel, 8 bytes
//     at row(2*oy + r) + (2*ox + s) * 8: overlapping windows of neighbouring output pixels cost nothing, every lane
//     supplies its own address.
// k' = r*32 + s*4 + c as in the forward kernel (14 blocks of 16; s = 7 and c = 3 are padding and are never written).
// Wavefront w: channel blocks {2(w&1), 2(w&1)+1} x kernel rows {0..3} (w < 2) or {4..6}; accumulators stay in registers
// over all (image, strip) items of the persistent workgroup; one fp32 slab [64][147] per workgroup, summed in a fixed
// order by pf_wrw_reduce (deterministic).
// =================================================================================================================
struct StemWrwArgs {
  const bf16_t* dY;  // [imgs][Ho][Wo][64]
  const bf16_t* X;   // [imgs][H][Wd][3]
  float* slabs;      // [S][64][147]
  int imgs, H, Wd, Ho, Wo;
  int strips, n_items;
  int rsb;           // LDS bytes per staged input row
  int dy_bytes;
};

#define SW_OROWS 2
#define SW_IROWS (2 * SW_OROWS + 5)

__global__ __launch_bounds__(ST_THREADS) void k_stem7x7_wrw(StemWrwArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;                                 // staged input rows [9][(Wd + 8)][4] bf16
  unsigned char* dys = smem + SW_IROWS * a.rsb;             // dY blocks [steps][4 pixel groups][4 channel blocks][8][16] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int nb0 = 2 * (wave & 1);                           // this wavefront's two 16-channel blocks
  const int r0 = (wave >> 1) ? 4 : 0, nr = (wave >> 1) ? 3 : 4;   // and its kernel rows
  const st_rsrc_t rsY = ST_MAKE_RSRC(a.dY, a.dy_bytes);
  constexpr uint32_t OOB = 0x80000000u;
  const int sb = lane >> 4, srow = (lane & 15) >> 1, sch = lane & 1;        // role inside one LDS-DMA instruction
  const int tr_off = (q & 1) * 128 + (l15 >> 2) * 32 + (l15 & 3) * 8;       // piece of the transposed read (k_wrw2)
  const int pg_lo = q >> 1;
  const int p_lo = 8 * pg_lo + 4 * (q & 1) + (l15 >> 2);                    // the pixel (within a 32-pixel step) whose X piece this lane addresses
  const int steps = (SW_OROWS * a.Wo) >> 5;
  const int pairs = a.Wd >> 1;

  f32x4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[i][k] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int t = tid; t < SW_IROWS * 8; t += ST_THREADS) {    // halo pixels, once
    const int row = t >> 3, h = t & 7;
    const int pp = h < 3 ? h : a.Wd + h;
    *reinterpret_cast<uint2*>(xs + (int64_t)row * a.rsb + pp * 8) = make_uint2(0u, 0u);
  }

  for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
    const int img = item / a.strips, strip = item - img * a.strips;
    const int oy0 = strip * SW_OROWS, iy0 = 2 * oy0 - 3;
    const int valid_px = ((a.Ho - oy0) < SW_OROWS ? (a.Ho - oy0) : SW_OROWS) * a.Wo;
    const int64_t m_base = ((int64_t)img * a.Ho + oy0) * a.Wo;
    __syncthreads();                                        // the previous item's reads are done
    // ---- dY: steps x 4 pixel groups, one LDS-DMA instruction (1 KB) each ----
    for (int id = wave; id < steps * 4; id += ST_THREADS / 64) {
      const int lin = id * 8 + srow;                        // id = step*4 + pg
      const uint32_t voff = (lin < valid_px) ? (uint32_t)(((m_base + lin) * ST_N + 16 * sb + 8 * sch) * 2) : OOB;
      ST_BUFFER_LOAD_LDS16(rsY, dys + id * 1024, voff);
    }
    // ---- X: 9 input rows, 3 -> 4 channels ----
    for (int row = tid >> 7; row < SW_IROWS; row += 2) {
      const int iy = iy0 + row;
      unsigned char* dst = xs + (int64_t)row * a.rsb + 3 * 8;
      const bool live = iy >= 0 && iy < a.H;
      const uint32_t* src = reinterpret_cast<const uint32_t*>(a.X + ((int64_t)img * a.H + (live ? iy : 0)) * a.Wd * 3);
      for (int j = tid & 127; j < pairs; j += 128) {
        uint32_t d0 = 0u, d1 = 0u, d2 = 0u;
        if (live) { d0 = src[3 * j]; d1 = src[3 * j + 1]; d2 = src[3 * j + 2]; }
        *reinterpret_cast<uint2*>(dst + j * 16) = make_uint2(d0, d1 & 0xFFFFu);
        *reinterpret_cast<uint2*>(dst + j * 16 + 8) = make_uint2((d1 >> 16) | (d2 << 16), d2 >> 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- 32 pixels per step ----
    for (int st = 0; st < steps; ++st) {
      const unsigned char* dbase = dys + st * 4096;
      bf16x8 df[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const st_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) st_v4s*)(dbase + (pg_lo * 4 + nb0 + i) * 256 + tr_off));
        const st_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) st_v4s*)(dbase + ((pg_lo + 2) * 4 + nb0 + i) * 256 + tr_off));
        st_v8s d8;
        d8[0] = lo[0]; d8[1] = lo[1]; d8[2] = lo[2]; d8[3] = lo[3];
        d8[4] = hi[0]; d8[5] = hi[1]; d8[6] = hi[2]; d8[7] = hi[3];
        df[i] = *reinterpret_cast<const bf16x8*>(&d8);
      }
      // this lane's two X pixels of the step: lin -> (output row of the strip, ox)
      int lin = st * 32 + p_lo;
      int orow = lin >= a.Wo ? 1 : 0;
      const unsigned char* xlo = xs + (int64_t)(2 * orow + r0) * a.rsb + (2 * (lin - orow * a.Wo) + (l15 & 3)) * 8;
      lin += 16;
      orow = lin >= a.Wo ? 1 : 0;
      const unsigned char* xhi = xs + (int64_t)(2 * orow + r0) * a.rsb + (2 * (lin - orow * a.Wo) + (l15 & 3)) * 8;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (rr < nr) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const st_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) st_v4s*)(xlo + (int64_t)rr * a.rsb + h * 32));
            const st_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) st_v4s*)(xhi + (int64_t)rr * a.rsb + h * 32));
            st_v8s x8;
            x8[0] = lo[0]; x8[1] = lo[1]; x8[2] = lo[2]; x8[3] = lo[3];
            x8[4] = hi[0]; x8[5] = hi[1]; x8[6] = hi[2]; x8[7] = hi[3];
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(&x8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[i][rr * 2 + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[i], xf, acc[i][rr * 2 + h], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- slab [64][7][7][3]: lane (q, l15) holds rows n = 16*nb + 4q + j, column k' = (r, s = 4h + l15/4, c = l15%4) ----
  float* slab = a.slabs + (int64_t)blockIdx.x * (ST_N * 147);
  const int c = l15 & 3;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    if (rr < nr) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int s = 4 * h + (l15 >> 2);
        if (s < 7 && c < 3) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              slab[(16 * (nb0 + i) + 4 * q + j) * 147 + ((r0 + rr) * 7 + s) * 3 + c] = acc[i][rr * 2 + h][j];
        }
      }
    }
  }
}

int pf_wrw_reduce(float* workspace, int S, int64_t n, void* dW, int dw_dtype, hipStream_t st);   // pf_conv.hip

static int stem_wrw_grid(int imgs, int H) {
  const int items = imgs * ((H / 2 + SW_OROWS - 1) / SW_OROWS);
  return items < 768 ? items : 768;
}

// number of fp32 slabs pf_conv_stem_wrw writes; the workspace must hold (slabs + 32) * 64 * 147 floats (0: unsupported)
extern "C" int pf_conv_stem_wrw_slabs(int imgs, int H, int Wd) {
  if (!pf_conv_stem_supported(H, Wd, 3, ST_N, 7, 2, 3) || imgs <= 0 || Wd > 256) return 0;
  if ((int64_t)imgs * (H / 2) * (Wd / 2) * ST_N * 2 >= (int64_t)1 << 31) return 0;
  return stem_wrw_grid(imgs, H);
}

extern "C" int pf_conv_stem_wrw(const void* dY, const void* X, void* dW, int dw_dtype, float* workspace, int imgs, int H,
                                int Wd, void* stream) {
  const int S = pf_conv_stem_wrw_slabs(imgs, H, Wd);
  if (S <= 0 || workspace == nullptr) return (int)hipErrorInvalidValue;
  if (!pf_aligned16(dY) || (reinterpret_cast<uintptr_t>(X) & 3u)) return (int)hipErrorInvalidValue;
  StemWrwArgs a;
  a.dY = (const bf16_t*)dY; a.X = (const bf16_t*)X; a.slabs = workspace;
  a.imgs = imgs; a.H = H; a.Wd = Wd; a.Ho = H / 2; a.Wo = Wd / 2;
  a.strips = (a.Ho + SW_OROWS - 1) / SW_OROWS;
  a.n_items = imgs * a.strips;
  a.rsb = (Wd + 8) * 8;
  a.dy_bytes = (int)((int64_t)imgs * a.Ho * a.Wo * ST_N * 2);
  const size_t lds = (size_t)SW_IROWS * a.rsb + (size_t)SW_OROWS * a.Wo * ST_N * 2;
  if (int e = pf_require_lds(reinterpret_cast<const void*>(&k_stem7x7_wrw), lds)) return e;
  k_stem7x7_wrw<<<S, ST_THREADS, lds, (hipStream_t)stream>>>(a);
  PF_LAUNCH_CHECK();
  return pf_wrw_reduce(workspace, S, (int64_t)ST_N * 147, dW, dw_dtype, (hipStream_t)stream);
}
