// The MobileNet-v1 stem: 3x3 / stride 2 convolution of a 3-channel image to 16 or 32 channels, TensorFlow 'SAME' padding
// (reference utils/external/mobilenet_v1.py:233-262, the first `slim.conv2d` of `mobilenet_v1_base`, conv_defs[0] = Conv(kernel=[3, 3],
// stride=2, depth=32), depth multipliers 0.5 / 1.0), forward and backward-filter, NHWC bf16.
//
// Why its own kernels (round 5).  Until now this layer ran on the general vector-ALU kernel of pf_convg.hip over a zero-padded COPY of
// the image: 361 us forward + 878 us backward-filter per step at 256 x 224 x 224 (profiles/r05_depthwise_layers_after.txt) against an
// HBM floor of ~63 us each and against MIOpen's 148 / 119 us -- 9 % of the MobileNet step, and an own kernel slower than the library
// it replaced.  The layout decisions are the ones of the ResNet stem (pf_stem.hip), re-derived for a 3 x 3 window:
//   * a workgroup stages the input rows of its output strip ONCE in LDS with every pixel padded from 3 to 4 channels (8 bytes), the
//     image's first pixel at LDS pixel `pw` (the front pad) and zeros around it -- the asymmetric 'SAME' padding of an even-sized image
//     (0 in front, 1 behind) needs no padded copy of the image: rows / columns outside it are staged as zeros;
//   * forward: k' = s * 4 + c per kernel row r, one MFMA k-step (32 values) per row with the taps s = 3 .. 7 and the channel c = 3 carrying
//     zero WEIGHTS: the eight k-values a lane feeds to v_mfma_f32_16x16x32_bf16 -- taps (2g, 2g + 1) x 4 channels -- are ONE aligned 16-byte
//     LDS read at row(2 oy + r) + (2 ox + 2 g) * 8 B.  27 of 96 k-values are real; the matrix work is 3 MFMAs per 16 pixels x 16 channels,
//     nothing next to the 282 MB the launch moves;
//   * weights are operand A, packed once per workgroup and kept in registers; the MFMA row -> channel map gives a lane 8 (N = 32) or 4
//     (N = 16) CONSECUTIVE channels of its pixel: one 16- / 8-byte store, no LDS transposition;
//   * backward-filter: both operands are pixel-major and the contraction runs over pixels, so both are fetched with the transposing LDS
//     read ds_read_b64_tr_b16 exactly as in k_stem7x7_wrw -- dY in [8 pixels][16 channels] blocks filled by LDS-DMA, X pieces = one
//     padded pixel each; float32 slabs [N][27] per workgroup, summed in a fixed order by pf_wrw_reduce (deterministic).
#include "pf_conv_common.h"

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef short s3_v4s __attribute__((ext_vector_type(4)));
typedef short s3_v8s __attribute__((ext_vector_type(8)));
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t s3_rsrc_t;
#define S3_MAKE_RSRC(p, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(p), (short)0, (int)(bytes), 0x00020000)
#define S3_BUFFER_LOAD_LDS16(rs, lds, voff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, voff, 0, 0, 0)
#else
typedef int s3_rsrc_t;
#define S3_MAKE_RSRC(p, bytes) 0
#define S3_BUFFER_LOAD_LDS16(rs, lds, voff) ((void)(rs), (void)(lds), (void)(voff))
#endif

#define S3_THREADS 256
#define S3_OROWS 8                      // forward: output rows per item
#define S3_IROWS (2 * S3_OROWS + 1)     // input rows staged per item
#define S3_R 3

struct Stem3Args {
  const bf16_t* X;   // [imgs][H][Wd][3]
  const bf16_t* W;   // [N][3][3][3]
  bf16_t* Y;         // [imgs][Ho][Wo][N]
  int imgs, H, Wd, Ho, Wo, ph, pw;
  int strips, n_items;
  int rsb;           // LDS bytes per staged input row = (Wd + 8) * 8
};

// one staged input row: Wd pixels of 3 channels -> 4 channels at LDS pixel pw.. (the halo pixels are zeroed once per workgroup)
__device__ __forceinline__ void s3_stage_row(unsigned char* dst, const bf16_t* X, int64_t row_elems_off, bool live, int pairs, int t0, int tstep) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(X + (live ? row_elems_off : 0));
  for (int j = t0; j < pairs; j += tstep) {         // two pixels (12 bytes) per task
    uint32_t d0 = 0u, d1 = 0u, d2 = 0u;
    if (live) { d0 = src[3 * j]; d1 = src[3 * j + 1]; d2 = src[3 * j + 2]; }
    *reinterpret_cast<uint2*>(dst + j * 16) = make_uint2(d0, d1 & 0xFFFFu);
    *reinterpret_cast<uint2*>(dst + j * 16 + 8) = make_uint2((d1 >> 16) | (d2 << 16), d2 >> 16);
  }
}

template <int NB>
__global__ __launch_bounds__(S3_THREADS) void k_stem3x3_fwd(Stem3Args a) {
  constexpr int N = 16 * NB;
  constexpr int WL_BYTES = N * S3_R * 32 * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;                               // packed weights [N][3][32] bf16
  unsigned char* xs = smem + WL_BYTES;                    // staged rows [17][(Wd + 8)][4] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;

  // ---- pack the weights: chunk (n, r, q) = taps (2q, 2q+1) x 4 channels of kernel row r of output channel n ----
  for (int ch = tid; ch < N * S3_R * 4; ch += S3_THREADS) {
    const int q = ch & 3, nr = ch >> 2;                   // nr = n*3 + r
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    if (q < 2) {
      const bf16_t* src = a.W + ((int64_t)nr * 3 + 2 * q) * 3;
      const uint32_t e0 = src[0], e1 = src[1], e2 = src[2];
      v[0] = e0 | (e1 << 16); v[1] = e2;
      if (q == 0) {                                       // tap 1 exists, tap 3 does not
        const uint32_t f0 = src[3], f1 = src[4], f2 = src[5];
        v[2] = f0 | (f1 << 16); v[3] = f2;
      }
    }
    *reinterpret_cast<uint4*>(wl + ((int64_t)nr * 32 + q * 8) * 2) = make_uint4(v[0], v[1], v[2], v[3]);
  }
  // ---- zero the halo pixels of every staged row once: LDS pixels [0, pw) and [pw + Wd, Wd + 8) ----
  for (int t = tid; t < S3_IROWS * 8; t += S3_THREADS) {
    const int row = t >> 3, h = t & 7;
    const int pp = h < a.pw ? h : a.Wd + h;
    *reinterpret_cast<uint2*>(xs + (int64_t)row * a.rsb + pp * 8) = make_uint2(0u, 0u);
  }
  __syncthreads();
  // ---- A fragments: MFMA row i of channel block nb is output channel 8*(i/4) + 4*nb + (i%4) (two blocks) or i (one) ----
  bf16x8 wf[NB][S3_R];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = (NB == 2) ? (8 * (l16 >> 2) + 4 * nb + (l16 & 3)) : l16;
#pragma unroll
    for (int r = 0; r < S3_R; ++r)
      wf[nb][r] = *reinterpret_cast<const bf16x8*>(wl + ((n * S3_R + r) * 32 + g * 8) * 2);
  }

  const int pairs = a.Wd >> 1;
  const int pblocks = a.Wo >> 4;                          // 16-pixel blocks per output row
  const int units = S3_OROWS * pblocks;
  for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
    const int img = item / a.strips, strip = item - img * a.strips;
    const int oy0 = strip * S3_OROWS, iy0 = 2 * oy0 - a.ph;
    __syncthreads();                                      // the previous item's fragment reads are done
    for (int row = tid >> 7; row < S3_IROWS; row += 2) {
      const int iy = iy0 + row;
      s3_stage_row(xs + (int64_t)row * a.rsb + a.pw * 8, a.X, ((int64_t)img * a.H + iy) * a.Wd * 3, iy >= 0 && iy < a.H, pairs, tid & 127, 128);
    }
    __syncthreads();
    for (int u = wave; u < units; u += S3_THREADS / 64) {
      const int orow = u / pblocks, pb = u - orow * pblocks;
      const int oy = oy0 + orow;
      if (oy >= a.Ho) continue;
      const int ox = pb * 16 + l16;
      const unsigned char* base = xs + (int64_t)(2 * orow) * a.rsb + (2 * ox + 2 * g) * 8;
      f32x4 acc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < S3_R; ++r) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(base + (int64_t)r * a.rsb);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][r], xf, acc[nb], 0, 0, 0);
      }
      bf16_t* out = a.Y + (((int64_t)img * a.Ho + oy) * a.Wo + ox) * N;
      if constexpr (NB == 2) {                            // lane: pixel ox, channels 8g + 4nb + j
        *reinterpret_cast<uint4*>(out + 8 * g) = make_uint4(pack_bf16x2(acc[0][0], acc[0][1]), pack_bf16x2(acc[0][2], acc[0][3]),
                                                            pack_bf16x2(acc[1][0], acc[1][1]), pack_bf16x2(acc[1][2], acc[1][3]));
      } else {                                            // channels 4g + j
        *reinterpret_cast<uint2*>(out + 4 * g) = make_uint2(pack_bf16x2(acc[0][0], acc[0][1]), pack_bf16x2(acc[0][2], acc[0][3]));
      }
    }
  }
}

static bool s3_shape_ok(int H, int Wd, int C, int N, int k, int stride, int ph, int pw, int Ho, int Wo) {
  return C == 3 && (N == 16 || N == 32) && k == 3 && stride == 2 && H > 0 && Wd >= 32 && (Wd % 2) == 0 && Wd <= 1024 && ph >= 0 && ph <= 1 &&
         pw >= 0 && pw <= 1 && Wo > 0 && (Wo % 16) == 0 && Ho > 0 &&
         // every window lies inside the staged rows / the zero halo: 2 (Wo - 1) + 7 <= Wd + 7, 2 (Ho - 1) + 2 - ph rows below H read zeros
         2 * Wo <= Wd + 2 && 2 * (Ho - 1) - ph < H;
}

// 1 when pf_conv_stem3_fwd / _wrw take the shape: 3 -> 16 | 32 channels, 3x3, stride 2, FRONT pads pad_h / pad_w in {0, 1} (positions
// behind the image read zeros whatever Ho / Wo say), Wo % 16 == 0
extern "C" int pf_conv_stem3_supported(int H, int Wd, int C, int N, int k, int stride, int pad_h, int pad_w, int Ho, int Wo) {
  return s3_shape_ok(H, Wd, C, N, k, stride, pad_h, pad_w, Ho, Wo) ? 1 : 0;
}

extern "C" int pf_conv_stem3_fwd(const void* X, const void* W, void* Y, int imgs, int H, int Wd, int N, int pad_h, int pad_w, int Ho,
                                 int Wo, void* stream) {
  if (!s3_shape_ok(H, Wd, 3, N, 3, 2, pad_h, pad_w, Ho, Wo) || imgs <= 0) return (int)hipErrorInvalidValue;
  if (!pf_aligned16(Y) || (reinterpret_cast<uintptr_t>(X) & 3u) || (reinterpret_cast<uintptr_t>(W) & 1u)) return (int)hipErrorInvalidValue;
  Stem3Args a;
  a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.Y = (bf16_t*)Y;
  a.imgs = imgs; a.H = H; a.Wd = Wd; a.Ho = Ho; a.Wo = Wo; a.ph = pad_h; a.pw = pad_w;
  a.strips = (Ho + S3_OROWS - 1) / S3_OROWS;
  a.n_items = imgs * a.strips;
  a.rsb = (Wd + 8) * 8;
  const size_t lds = (size_t)N * S3_R * 64 + (size_t)S3_IROWS * a.rsb + 16;
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  const void* fn = (N == 32) ? reinterpret_cast<const void*>(&k_stem3x3_fwd<2>) : reinterpret_cast<const void*>(&k_stem3x3_fwd<1>);
  if (int e = pf_require_lds(fn, lds)) return e;
  const int grid = a.n_items < 1024 ? a.n_items : 1024;
  if (N == 32) k_stem3x3_fwd<2><<<grid, S3_THREADS, lds, (hipStream_t)stream>>>(a);
  else k_stem3x3_fwd<1><<<grid, S3_THREADS, lds, (hipStream_t)stream>>>(a);
  PF_LAUNCH_CHECK();
  return 0;
}

// =================================================================================================================
// Backward-filter:  dW[n][r][s][c] = sum over (img, oy, ox) of dY[img][oy][ox][n] * X[img][2oy + r - ph][2ox + s - pw][c]
// Wavefront w: channel block w % NB, kernel rows {0, 1} / {2} (N = 32) or row w (N = 16; the fourth wavefront only stages).
// =================================================================================================================
struct Stem3WrwArgs {
  const bf16_t* dY;  // [imgs][Ho][Wo][N]
  const bf16_t* X;   // [imgs][H][Wd][3]
  float* slabs;      // [S][N][27]
  int imgs, H, Wd, Ho, Wo, ph, pw;
  int strips, n_items;
  int rsb;
  int dy_bytes;
};

#define S3W_OROWS 2
#define S3W_IROWS (2 * S3W_OROWS + 1)

template <int NB>
__global__ __launch_bounds__(S3_THREADS) void k_stem3x3_wrw(Stem3WrwArgs a) {
  constexpr int N = 16 * NB;
  constexpr int PGI = 4 / NB;                               // pixel groups (of 8) per LDS-DMA instruction
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;                                 // staged input rows [5][(Wd + 8)][4] bf16
  unsigned char* dys = smem + S3W_IROWS * a.rsb;            // dY blocks [steps][4 pixel groups][NB channel blocks][8][16] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int nb0 = (NB == 2) ? (wave & 1) : 0;               // this wavefront's 16-channel block
  const int r0 = (NB == 2) ? ((wave >> 1) ? 2 : 0) : wave;  // and its kernel rows r0 .. r0 + nr - 1
  const int nr = (NB == 2) ? ((wave >> 1) ? 1 : 2) : (wave < 3 ? 1 : 0);
  const s3_rsrc_t rsY = S3_MAKE_RSRC(a.dY, a.dy_bytes);
  constexpr uint32_t OOB = 0x80000000u;
  // role inside one LDS-DMA instruction (1 KiB = PGI pixel groups x NB channel blocks x [8 pixels][16 channels])
  const int spg = lane / (16 * NB), sb = (lane >> 4) % NB, srow = (lane & 15) >> 1, sch = lane & 1;
  const int tr_off = (q & 1) * 128 + (l15 >> 2) * 32 + (l15 & 3) * 8;       // piece of the transposed read (as in k_wrw2 / k_stem7x7_wrw)
  const int pg_lo = q >> 1;
  const int p_lo = 8 * pg_lo + 4 * (q & 1) + (l15 >> 2);                    // the pixel (within a 32-pixel step) whose X piece this lane addresses
  const int steps = (S3W_OROWS * a.Wo) >> 5;
  const int pairs = a.Wd >> 1;

  f32x4 acc[2];
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];

  for (int t = tid; t < S3W_IROWS * 8; t += S3_THREADS) {   // halo pixels, once
    const int row = t >> 3, h = t & 7;
    const int pp = h < a.pw ? h : a.Wd + h;
    *reinterpret_cast<uint2*>(xs + (int64_t)row * a.rsb + pp * 8) = make_uint2(0u, 0u);
  }

  for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
    const int img = item / a.strips, strip = item - img * a.strips;
    const int oy0 = strip * S3W_OROWS, iy0 = 2 * oy0 - a.ph;
    const int valid_px = ((a.Ho - oy0) < S3W_OROWS ? (a.Ho - oy0) : S3W_OROWS) * a.Wo;
    const int64_t m_base = ((int64_t)img * a.Ho + oy0) * a.Wo;
    __syncthreads();                                        // the previous item's reads are done
    // ---- dY: steps x 4 pixel groups x NB channel blocks, one LDS-DMA instruction per KiB ----
    for (int id = wave; id < steps * NB; id += S3_THREADS / 64) {
      const int lin = (id * PGI + spg) * 8 + srow;
      const uint32_t voff = (lin < valid_px) ? (uint32_t)(((m_base + lin) * N + 16 * sb + 8 * sch) * 2) : OOB;
      S3_BUFFER_LOAD_LDS16(rsY, dys + id * 1024, voff);
    }
    // ---- X: 5 input rows, 3 -> 4 channels ----
    for (int row = tid >> 7; row < S3W_IROWS; row += 2) {
      const int iy = iy0 + row;
      s3_stage_row(xs + (int64_t)row * a.rsb + a.pw * 8, a.X, ((int64_t)img * a.H + iy) * a.Wd * 3, iy >= 0 && iy < a.H, pairs, tid & 127, 128);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- 32 pixels per step ----
    for (int st = 0; st < steps; ++st) {
      const unsigned char* dbase = dys + st * (4 * NB * 256);
      const s3_v4s dlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s3_v4s*)(dbase + (pg_lo * NB + nb0) * 256 + tr_off));
      const s3_v4s dhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s3_v4s*)(dbase + ((pg_lo + 2) * NB + nb0) * 256 + tr_off));
      s3_v8s d8;
      d8[0] = dlo[0]; d8[1] = dlo[1]; d8[2] = dlo[2]; d8[3] = dlo[3];
      d8[4] = dhi[0]; d8[5] = dhi[1]; d8[6] = dhi[2]; d8[7] = dhi[3];
      const bf16x8 df = *reinterpret_cast<const bf16x8*>(&d8);
      // this lane's two X pixels of the step: lin -> (output row of the strip, ox)
      int lin = st * 32 + p_lo;
      int orow = lin >= a.Wo ? 1 : 0;
      const unsigned char* xlo = xs + (int64_t)(2 * orow + r0) * a.rsb + (2 * (lin - orow * a.Wo) + (l15 & 3)) * 8;
      lin += 16;
      orow = lin >= a.Wo ? 1 : 0;
      const unsigned char* xhi = xs + (int64_t)(2 * orow + r0) * a.rsb + (2 * (lin - orow * a.Wo) + (l15 & 3)) * 8;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        if (rr < nr) {
          const s3_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s3_v4s*)(xlo + (int64_t)rr * a.rsb));
          const s3_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s3_v4s*)(xhi + (int64_t)rr * a.rsb));
          s3_v8s x8;
          x8[0] = lo[0]; x8[1] = lo[1]; x8[2] = lo[2]; x8[3] = lo[3];
          x8[4] = hi[0]; x8[5] = hi[1]; x8[6] = hi[2]; x8[7] = hi[3];
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(&x8);
          acc[rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, xf, acc[rr], 0, 0, 0);
        }
      }
    }
  }
  // ---- slab [N][3][3][3]: lane (q, l15) holds rows n = 16*nb + 4q + j, column k' = (s = l15/4, c = l15%4) of kernel row r0 + rr ----
  float* slab = a.slabs + (int64_t)blockIdx.x * (N * 27);
  const int s = l15 >> 2, c = l15 & 3;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    if (rr < nr && s < 3 && c < 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) slab[(16 * nb0 + 4 * q + j) * 27 + ((r0 + rr) * 3 + s) * 3 + c] = acc[rr][j];
    }
  }
}

int pf_wrw_reduce(float* workspace, int S, int64_t n, void* dW, int dw_dtype, hipStream_t st);   // pf_conv.hip

static int s3_wrw_grid(int imgs, int Ho) {
  const int items = imgs * ((Ho + S3W_OROWS - 1) / S3W_OROWS);
  return items < 768 ? items : 768;
}

// number of fp32 slabs pf_conv_stem3_wrw writes; the workspace must hold (slabs + 32) * N * 27 floats (0: unsupported)
extern "C" int pf_conv_stem3_wrw_slabs(int imgs, int H, int Wd, int N, int pad_h, int pad_w, int Ho, int Wo) {
  if (!s3_shape_ok(H, Wd, 3, N, 3, 2, pad_h, pad_w, Ho, Wo) || imgs <= 0 || Wd > 512) return 0;
  if ((int64_t)imgs * Ho * Wo * N * 2 >= (int64_t)1 << 31) return 0;
  return s3_wrw_grid(imgs, Ho);
}

extern "C" int pf_conv_stem3_wrw(const void* dY, const void* X, void* dW, int dw_dtype, float* workspace, int imgs, int H, int Wd, int N,
                                 int pad_h, int pad_w, int Ho, int Wo, void* stream) {
  const int S = pf_conv_stem3_wrw_slabs(imgs, H, Wd, N, pad_h, pad_w, Ho, Wo);
  if (S <= 0 || workspace == nullptr) return (int)hipErrorInvalidValue;
  if (!pf_aligned16(dY) || (reinterpret_cast<uintptr_t>(X) & 3u)) return (int)hipErrorInvalidValue;
  Stem3WrwArgs a;
  a.dY = (const bf16_t*)dY; a.X = (const bf16_t*)X; a.slabs = workspace;
  a.imgs = imgs; a.H = H; a.Wd = Wd; a.Ho = Ho; a.Wo = Wo; a.ph = pad_h; a.pw = pad_w;
  a.strips = (Ho + S3W_OROWS - 1) / S3W_OROWS;
  a.n_items = imgs * a.strips;
  a.rsb = (Wd + 8) * 8;
  a.dy_bytes = (int)((int64_t)imgs * Ho * Wo * N * 2);
  const size_t lds = (size_t)S3W_IROWS * a.rsb + (size_t)S3W_OROWS * Wo * N * 2 + 16;
  const void* fn = (N == 32) ? reinterpret_cast<const void*>(&k_stem3x3_wrw<2>) : reinterpret_cast<const void*>(&k_stem3x3_wrw<1>);
  if (int e = pf_require_lds(fn, lds)) return e;
  if (N == 32) k_stem3x3_wrw<2><<<S, S3_THREADS, lds, (hipStream_t)stream>>>(a);
  else k_stem3x3_wrw<1><<<S, S3_THREADS, lds, (hipStream_t)stream>>>(a);
  PF_LAUNCH_CHECK();
  return pf_wrw_reduce(workspace, S, (int64_t)N * 27, dW, dw_dtype, (hipStream_t)stream);
}
