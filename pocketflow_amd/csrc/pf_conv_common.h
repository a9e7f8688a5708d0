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
  // output affine (oss != null): the row pass stores act(scale[n] * y + shift[n]) instead of y -- the inference-mode BN + activation
  // of the CONSUMER folded into this launch (out_affine8 below); no statistics, no residual with it
  const float* oss;     // scale | shift   [2][N]
  int oact;             // PF_ACT_*
};

// The output affine on one 16-byte vector of the row pass: 8 consecutive channels n .. n + 7 of one pixel, ALREADY rounded to bf16.
// Exactly the arithmetic of pf_bn_act_quant_apply without a quantiser on exactly the value it would read back from HBM --
// act(fma(scale, y, shift)), rounded to bf16 once -- so folding the pass into the producing convolution changes no bit.
__device__ __forceinline__ uint4 out_affine8(const uint4& c, const float* __restrict__ oss, int N, int n, int act) {
  // (called from kernels instantiated with AFF only: the four constant loads are loop-invariant in a row pass and the compiler may
  // hoist them -- re-issued per 16-byte vector they cost the teacher's launches 45 us each, profiles/r06_step_kernels_b256.csv)
  float f[8];
  unpack8(c, f);
  const float4 s0 = *reinterpret_cast<const float4*>(oss + n), s1 = *reinterpret_cast<const float4*>(oss + n + 4);
  const float4 h0 = *reinterpret_cast<const float4*>(oss + N + n), h1 = *reinterpret_cast<const float4*>(oss + N + n + 4);
  const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float y = fmaf(sc[j], f[j], sh[j]);
    if (act != PF_ACT_NONE) y = fmaxf(y, 0.f);
    if (act == PF_ACT_RELU6) y = fminf(y, 6.f);
    f[j] = y;
  }
  return pack8(f);
}

__device__ __forceinline__ int64_t map_row(const ConvArgs& a, int m) {
  if (a.stride == 1) return m;
  const int hw = a.Ho * a.Wo;
  const int n = m / hw, rem = m - n * hw;
  const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
  return ((int64_t)n * a.H + (int64_t)ho * a.stride) * a.Wd + (int64_t)wo * a.stride;
}

// prologue on one 16-byte vector (8 consecutive input channels of one pixel): BN affine -> activation clamp ->
// activation fake-quant of the producer layer (uq utils.py:51-79,163-199), bf16 throughput mode.
//
// All per-tensor and per-channel constants are FOLDED once (pro_fold_*), so that an element costs
//   t = med3(fma(sc', x, sh'), lo', hi');  q = fma(rint(t), alpha/k, beta)          (4 VALU + unpack / pack)
// with sc' = scale * k/alpha, sh' = (shift - beta) * k/alpha, lo' = (lo - beta) * k/alpha, hi' likewise: t is the grid
// coordinate (y - beta) * k/alpha of the reference chain with the BN affine and the quantiser affine contracted into
// ONE fma (round 2 spent 7 VALU here: fma, max, min, sub, mul, rint, fma).  t differs from the chain's value by a few
// float32 ulps of its terms, so q differs from the five-rounding chain of uq_point() only where t sits within that
// distance of a rounding boundary n + 1/2 -- enumerated by tests/test_conv_gpu.py (tie test).  Without fake-quant
// (quant == 0) the fold is the identity (k/alpha = 1, beta = 0): t = act(fma(scale, x, shift)) exactly as before.
struct Pro {
  float sc[8], sh[8];   // FOLDED scale / shift of this lane's 8 channels (pro_fold_scale / pro_fold_shift)
  float lo, hi;         // FOLDED activation window (pro_fold_window)
  float beta, c1, c2;   // range minimum, k/alpha, alpha/k
  int quant;
};
// a: activation window (act_lo, act_hi), kq = 2^bits - 1, slot = min/max slot of the activation or null
__device__ __forceinline__ void pro_init(Pro& p, float act_lo, float act_hi, float kq, const uint32_t* slot) {
  p.quant = 0; p.beta = 0.f; p.c1 = 1.f; p.c2 = 1.f;
  if (slot != nullptr) {
    float alpha, beta;
    slot_alpha_beta(slot, alpha, beta);
    p.quant = 1; p.beta = beta; p.c1 = kq / alpha; p.c2 = alpha / kq;
  }
  p.lo = (act_lo - p.beta) * p.c1;
  p.hi = (act_hi - p.beta) * p.c1;
}
__device__ __forceinline__ float pro_fold_scale(const Pro& p, float scale) { return scale * p.c1; }
__device__ __forceinline__ float pro_fold_shift(const Pro& p, float shift) { return (shift - p.beta) * p.c1; }
__device__ __forceinline__ float pro_point(const Pro& p, float sc, float sh, float x) {
  const float t = __builtin_amdgcn_fmed3f(fmaf(sc, x, sh), p.lo, p.hi);
  return p.quant ? fmaf(rintf(t), p.c2, p.beta) : t;
}
// element by element: for kernels at their register limit (the packed form below needs aligned register pairs; the 128-wide
// resident kernel spilled 36-48 bytes with it and lost 25 %)
__device__ __forceinline__ uint4 pro_apply_scalar(const Pro& p, const uint4& v) {
  float f[8];
  unpack8(v, f);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = pro_point(p, p.sc[j], p.sh[j], f[j]);
  return pack8(f);
}
// Eight elements of one 16-byte vector.  Same arithmetic as pro_point element by element (bit-identical results), arranged for
// the hardware: the two fused multiply-adds run as v_pk_fma_f32 on element PAIRS (two fp32 FMAs per lane and issue), and the
// quantising / non-quantising variants are two loops behind ONE wave-uniform branch instead of a per-element select
// (6.5 -> 4.5 VALU per element with the unpack and the packed conversion).
typedef float f32x2p_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 pro_apply(const Pro& p, const uint4& v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t o[4];
  if (p.quant) {
    const f32x2p_t c2 = {p.c2, p.c2}, be = {p.beta, p.beta};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2p_t x = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xFFFF0000u)};
      const f32x2p_t sc = {p.sc[2 * i], p.sc[2 * i + 1]}, sh = {p.sh[2 * i], p.sh[2 * i + 1]};
      const f32x2p_t t = __builtin_elementwise_fma(sc, x, sh);
      const f32x2p_t r = {rintf(__builtin_amdgcn_fmed3f(t[0], p.lo, p.hi)), rintf(__builtin_amdgcn_fmed3f(t[1], p.lo, p.hi))};
      const f32x2p_t q = __builtin_elementwise_fma(r, c2, be);
      o[i] = pack_bf16x2(q[0], q[1]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2p_t x = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xFFFF0000u)};
      const f32x2p_t sc = {p.sc[2 * i], p.sc[2 * i + 1]}, sh = {p.sh[2 * i], p.sh[2 * i + 1]};
      const f32x2p_t t = __builtin_elementwise_fma(sc, x, sh);
      o[i] = pack_bf16x2(__builtin_amdgcn_fmed3f(t[0], p.lo, p.hi), __builtin_amdgcn_fmed3f(t[1], p.lo, p.hi));
    }
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// ---- LDS accesses the compiler must NOT order against LDS-DMA ------------------------------------------------------
// hipcc (ROCm 7.2) treats a pending `buffer_load ... lds` as a store to LDS that any ordinary LDS store -- and any LDS load
// it cannot tell apart -- may alias, and emits `s_waitcnt vmcnt(0)` in front of it: one such access per k-step drains a
// multi-stage LDS-DMA ring every step (seen in the ISA of the 3-stage prologue kernel: the wait sat in front of the
// in-place ds_write of the prologue pass and in front of the scale / shift reads).  These helpers issue the access from
// inline asm, which the wait-count pass does not model; the CALLER guarantees the ordering (own counted vmcnt for the
// vectors a lane staged itself; s_waitcnt lgkmcnt(0) + barrier before anybody else reads what was written here).
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));
// loads and their wait in ONE statement (outputs early-clobber): the destinations are valid when the statement ends
__device__ __forceinline__ void lds_read_b128x4(uint32_t p0, uint32_t p1, float4& a, float4& b, float4& c, float4& d) {
  f32x4v_t ra, rb, rc, rd;
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:16\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(ra), "=&v"(rb), "=&v"(rc), "=&v"(rd) : "v"(p0), "v"(p1) : "memory");
  a = make_float4(ra[0], ra[1], ra[2], ra[3]); b = make_float4(rb[0], rb[1], rb[2], rb[3]);
  c = make_float4(rc[0], rc[1], rc[2], rc[3]); d = make_float4(rd[0], rd[1], rd[2], rd[3]);
}
// the four constant vectors of a prologue step AND one / two data vectors behind ONE wait
__device__ __forceinline__ void lds_read_b128x4_1(uint32_t p0, uint32_t p1, uint32_t pv, float4& a, float4& b, float4& c, float4& d, uint4& v) {
  f32x4v_t ra, rb, rc, rd;
  u32x4_t rv;
  asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:16\n\t"
               "ds_read_b128 %4, %7\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(ra), "=&v"(rb), "=&v"(rc), "=&v"(rd), "=&v"(rv) : "v"(p0), "v"(p1), "v"(pv) : "memory");
  a = make_float4(ra[0], ra[1], ra[2], ra[3]); b = make_float4(rb[0], rb[1], rb[2], rb[3]);
  c = make_float4(rc[0], rc[1], rc[2], rc[3]); d = make_float4(rd[0], rd[1], rd[2], rd[3]);
  v = make_uint4(rv[0], rv[1], rv[2], rv[3]);
}
__device__ __forceinline__ void lds_read_b128x4_2(uint32_t p0, uint32_t p1, uint32_t pv0, uint32_t pv1, float4& a, float4& b, float4& c,
                                                  float4& d, uint4& v0, uint4& v1) {
  f32x4v_t ra, rb, rc, rd;
  u32x4_t r0, r1;
  asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:16\n\t"
               "ds_read_b128 %4, %8\n\tds_read_b128 %5, %9\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(ra), "=&v"(rb), "=&v"(rc), "=&v"(rd), "=&v"(r0), "=&v"(r1) : "v"(p0), "v"(p1), "v"(pv0), "v"(pv1) : "memory");
  a = make_float4(ra[0], ra[1], ra[2], ra[3]); b = make_float4(rb[0], rb[1], rb[2], rb[3]);
  c = make_float4(rc[0], rc[1], rc[2], rc[3]); d = make_float4(rd[0], rd[1], rd[2], rd[3]);
  v0 = make_uint4(r0[0], r0[1], r0[2], r0[3]); v1 = make_uint4(r1[0], r1[1], r1[2], r1[3]);
}
__device__ __forceinline__ uint4 lds_read_b128(uint32_t p) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_read_b128x2(uint32_t p0, uint32_t p1, uint4& a, uint4& b) {
  u32x4_t ra, rb;
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ra), "=&v"(rb) : "v"(p0), "v"(p1) : "memory");
  a = make_uint4(ra[0], ra[1], ra[2], ra[3]); b = make_uint4(rb[0], rb[1], rb[2], rb[3]);
}
// a read WITHOUT its wait, and the wait that makes a batch of them valid (the registers are tied to the wait statement, so no use
// of them can be scheduled in front of it)
__device__ __forceinline__ void lds_read_b128_nowait(uint32_t p, u32x4_t& v) {
  asm volatile("ds_read_b128 %0, %1" : "=&v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void lds_wait_batch8(u32x4_t (&v)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
}
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_write_b64(uint32_t p, const uint2& v) {
  const u32x2_t r = {v.x, v.y};
  asm volatile("ds_write_b64 %0, %1" : : "v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ void lds_write_b128(uint32_t p, const uint4& v) {
  const u32x4_t r = {v.x, v.y, v.z, v.w};
  asm volatile("ds_write_b128 %0, %1" : : "v"(p), "v"(r) : "memory");
}
