// Shared by the convolution kernels (pf_conv.hip, pf_conv_stream.hip, pf_igemm.hip): argument block, the
// BN/act/fake-quant prologue on a 16-byte vector, bf16 packing helpers.
#pragma once
#include "pf_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CV_BM 128
#define CV_BK 64
#define CV_LDK (CV_BK + 8)
#define CV_MAXK 2048     // largest input-channel count the prologue can stage in LDS

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  // one v_cvt_pk_bf16_f32 (round-to-nearest-even, same as f32_to_bf16)
  const f32x2_t v = {a, b};
  const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<const uint32_t*>(&r);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* o) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xFFFF0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xFFFF0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xFFFF0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float* o) {
  return make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]),
                    pack_bf16x2(o[6], o[7]));
}

// Everything the kernels need, passed by value.
struct ConvArgs {
  const bf16_t* X;      // [rows_in][K]   pre-BN producer output (or any NHWC tensor)
  const bf16_t* W;      // fwd: [N][K]    wrw: dY [M][N]
  bf16_t* Y;            // fwd: [M][N]
  const bf16_t* R;      // residual [M][N] or null
  const float* ss;      // prologue scale[K] | shift[K], or null
  const uint32_t* slot; // activation min/max slot (null: no fake-quant in the prologue)
  float* partial;       // fwd: stats [G][4][N] or null;  wrw: [S][N][K] fp32
  float kq, act_lo, act_hi;
  int M, N, K;          // output pixels, output channels, input channels
  int tiles_m, tiles_n, G;
  // strided 1x1: output pixel (n, ho, wo) reads input pixel (n, ho*stride, wo*stride); stride == 1: identity
  int Ho, Wo, H, Wd, stride;
  int ymap;             // fwd as backward-data of a strided conv: map the OUTPUT rows instead of the input rows
  int rows_per_split;   // wrw
  // backward-data with the BN-backward statistics of the consumer of dQ in the epilogue (bx != null):
  // partial[g][2][N] = per-channel {sum dy, sum dy * xhat}, dy = dq * act'(scale*x+shift), xhat = (x-mean)*invstd
  const bf16_t* bx;     // the BN's input x, [M][N]
  const float* bss;     // its scale | shift   [2][N]
  const float* bmi;     // its mean | invstd   [2][N]
  float b_lo, b_hi;     // activation window of the mask: lo < u < hi  (ReLU: 0, +inf; ReLU6: 0, 6)
};

__device__ __forceinline__ int64_t map_row(const ConvArgs& a, int m) {
  if (a.stride == 1) return m;
  const int hw = a.Ho * a.Wo;
  const int n = m / hw, rem = m - n * hw;
  const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
  return ((int64_t)n * a.H + (int64_t)ho * a.stride) * a.Wd + (int64_t)wo * a.stride;
}

// prologue on one 16-byte vector (8 consecutive input channels of one pixel)
struct Pro {
  float sc[8], sh[8];
  float lo, hi, beta, c1, c2;
  int quant;
};
__device__ __forceinline__ uint4 pro_apply(const Pro& p, const uint4& v) {
  float f[8];
  unpack8(v, f);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float y = fminf(fmaxf(fmaf(p.sc[j], f[j], p.sh[j]), p.lo), p.hi);
    // fake-quant with the per-tensor constants folded: rint((y - beta) * k/alpha) * alpha/k + beta.
    // Differs from the five-rounding chain of uq_point() only on exact rounding ties (bf16 throughput
    // mode; the float32 parity mode never takes this path).
    if (p.quant) y = fmaf(rintf((y - p.beta) * p.c1), p.c2, p.beta);
    f[j] = y;
  }
  return pack8(f);
}

