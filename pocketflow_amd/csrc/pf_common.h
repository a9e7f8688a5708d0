// Device-side helpers shared by the gfx950 kernels.  CDNA4 only: 64-lane wavefronts, no
// compatibility paths.  The whole library is compiled with -ffp-contract=off so that every
// float32 mul / add / div is rounded exactly once, like the reference's one-TF-op-per-rounding
// chains; fused multiply-adds are written explicitly (fmaf) where they are wanted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pocketflow_hip.h"

#define PF_WAVE 64
#define PF_THREADS 256
#define PF_MAX_GRID 2048   // 256 CUs x 8 workgroups: grid-stride beyond this

#define PF_LAUNCH_CHECK()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

typedef uint16_t bf16_t;

// hipFuncAttributeMaxDynamicSharedMemorySize of `fn` raised to at least `lds` bytes ON THE CURRENT DEVICE (the attribute is per
// device; launches that need more than 64 KiB of LDS fail without it).  Cached per (device, function) behind a mutex: a process that
// drives a second GPU, or launches first from another thread, is configured too (the caches were process-wide statics).
int pf_require_lds(const void* fn, size_t lds);

// ---- tuning / A-B switches (host) --------------------------------------------------------------
// Every PF_* environment switch of the launchers is read ONCE, on first use, into this struct (pf_api.hip): a launcher and the
// workspace-size query that precedes it always see the same decision, and no launch pays for getenv / atoi / sscanf.  Tools and
// tests that flip a switch in-process call pf_tuning_reload() (hip.tuning_reload()) afterwards.
struct PfTuning {
  int conv_bn;                    // PF_CONV_BN            0 | 64 | 128
  int conv_igemm, conv_igemm_pro; // PF_CONV_IGEMM[_PRO]   1 (default) | 0
  int conv_stream;                // PF_CONV_STREAM        1 (default) | 0
  int conv_stream_maxsplit;       // PF_CONV_STREAM_MAXSPLIT  2
  int igemm_pro3;                 // PF_IGEMM_PRO3         1 (default) | 0
  int igemm_tile_bm, igemm_tile_bn;   // PF_IGEMM_TILE "BMxBN" (0, 0: none)
  int pool3s2;                    // PF_POOL3S2            1 (default) | 0
  int wrw_tr, wrw2, wrw2_target;  // PF_WRW_TR 0, PF_WRW2 1, PF_WRW2_TARGET 0 (= per-shape default)
  int wrw2_tk256;                 // PF_WRW2_TK256         1 (default) | 0: 256-input-channel tiles of the shared-tile backward-filter kernel for N <= 128 (round 6)
  int splitk;                     // PF_IGEMM_SPLITK       1 (default) | 0
  int conv3x3_c64;                // PF_CONV3X3_C64        1 (default) | 0: the window-staged kernel for 3x3, 64 -> 64 channels, 56 x 56 (pf_conv3x3_c64.hip)
};
const PfTuning& pf_tuning();

// ---- bf16 <-> f32 (round-to-nearest-even, same as torch / XLA) ------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (bf16_t)((u >> 16) | 0x0040u);  // quiet NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// ---- 16-byte vector access: 4 x f32 or 8 x bf16 per lane -------------------------------------
template <typename T> struct VecTraits;
template <> struct VecTraits<float> { static constexpr int N = 4; };
template <> struct VecTraits<bf16_t> { static constexpr int N = 8; };

template <typename T>
__device__ __forceinline__ void load_vec(const T* __restrict__ p, float* out);
template <>
__device__ __forceinline__ void load_vec<float>(const float* __restrict__ p, float* out) {
  float4 v = *reinterpret_cast<const float4*>(p);
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <>
__device__ __forceinline__ void load_vec<bf16_t>(const bf16_t* __restrict__ p, float* out) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xFFFF0000u);
  out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xFFFF0000u);
  out[4] = __uint_as_float(v.z << 16); out[5] = __uint_as_float(v.z & 0xFFFF0000u);
  out[6] = __uint_as_float(v.w << 16); out[7] = __uint_as_float(v.w & 0xFFFF0000u);
}
template <typename T>
__device__ __forceinline__ void store_vec(T* __restrict__ p, const float* in);
template <>
__device__ __forceinline__ void store_vec<float>(float* __restrict__ p, const float* in) {
  *reinterpret_cast<float4*>(p) = make_float4(in[0], in[1], in[2], in[3]);
}
template <>
__device__ __forceinline__ void store_vec<bf16_t>(bf16_t* __restrict__ p, const float* in) {
  uint4 v;
  v.x = (uint32_t)f32_to_bf16(in[0]) | ((uint32_t)f32_to_bf16(in[1]) << 16);
  v.y = (uint32_t)f32_to_bf16(in[2]) | ((uint32_t)f32_to_bf16(in[3]) << 16);
  v.z = (uint32_t)f32_to_bf16(in[4]) | ((uint32_t)f32_to_bf16(in[5]) << 16);
  v.w = (uint32_t)f32_to_bf16(in[6]) | ((uint32_t)f32_to_bf16(in[7]) << 16);
  *reinterpret_cast<uint4*>(p) = v;
}
// 8 elements per lane per iteration for every dtype (f32: two 16-B accesses, bf16: one)
template <typename T> __device__ __forceinline__ void load8(const T* __restrict__ p, float* out);
template <> __device__ __forceinline__ void load8<float>(const float* __restrict__ p, float* out) {
  load_vec<float>(p, out); load_vec<float>(p + 4, out + 4);
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* __restrict__ p, float* out) {
  load_vec<bf16_t>(p, out);
}
template <typename T> __device__ __forceinline__ void store8(T* __restrict__ p, const float* in);
template <> __device__ __forceinline__ void store8<float>(float* __restrict__ p, const float* in) {
  store_vec<float>(p, in); store_vec<float>(p + 4, in + 4);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* __restrict__ p, const float* in) {
  store_vec<bf16_t>(p, in);
}
__host__ __device__ __forceinline__ bool pf_aligned16(const void* p) {
  return (((uintptr_t)p) & 15) == 0;
}
template <typename T> __device__ __forceinline__ float load_one(const T* p);
template <> __device__ __forceinline__ float load_one<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_one<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void store_one(T* p, float v);
template <> __device__ __forceinline__ void store_one<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_one<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// ---- order-preserving float <-> uint32 encoding for atomicMin-based min/max -----------------
__device__ __forceinline__ uint32_t enc_f32(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e) {
  uint32_t u = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
  return __uint_as_float(u);
}
// slot = {enc(min), ~enc(max)}; alpha = (max - min) + 1e-10f ; beta = min  (uq utils.py:226-229)
__device__ __forceinline__ void slot_alpha_beta(const uint32_t* slot, float& alpha, float& beta) {
  const uint2 sv = *reinterpret_cast<const uint2*>(slot);   // one 8-byte load (slots are 8-B aligned)
  float mn = dec_f32(sv.x);
  float mx = dec_f32(~sv.y);
  alpha = (mx - mn) + 1e-10f;
  beta = mn;
}

// ---- activation ------------------------------------------------------------------------------
// Running per-channel minimum / maximum of the statistics passes.  fminf / fmaxf compile to FIVE instructions per element here:
// the kernels run in IEEE mode, where v_min / v_max quiet a signalling NaN instead of ignoring it, so hipcc canonicalises every
// operand it cannot prove quiet (v_max x, x, x of the value -- unpacked with a shift -- and of BOTH running values, which are loop
// phis).  The accumulators therefore issue the bare instruction (round 4): identical for
// every input except a SIGNALLING NaN (result NaN instead of the other operand); quiet NaNs are ignored either way.
__device__ __forceinline__ float pf_acc_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float pf_acc_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int ACT> __device__ __forceinline__ float apply_act(float x) {
  if (ACT == PF_ACT_RELU) return fmaxf(x, 0.0f);
  if (ACT == PF_ACT_RELU6) return fminf(fmaxf(x, 0.0f), 6.0f);
  return x;
}
template <int ACT> __device__ __forceinline__ float act_mask(float u) {
  if (ACT == PF_ACT_RELU) return u > 0.0f ? 1.0f : 0.0f;
  if (ACT == PF_ACT_RELU6) return (u > 0.0f && u < 6.0f) ? 1.0f : 0.0f;
  return 1.0f;
}

// ---- the fake-quant point function (uq utils.py:182-187, 245): exactly five roundings --------
__device__ __forceinline__ float uq_point(float t, float alpha, float beta, float k) {
  float xn = (t - beta) / alpha;
  float q = rintf(xn * k) / k;       // rintf == round-half-to-even == tf.round
  return alpha * q + beta;           // two roundings (library built with -ffp-contract=off)
}
__host__ __device__ __forceinline__ float uq_k_of_bits(int bits) {
  // float32(2**bits - 1): 8 -> 255, 32 -> 4294967296.0f.  2^bits - 1 computed in float with one
  // round-to-nearest-even rounding equals the int64 -> float32 cast for every bits in 1..32
  // (and avoids a wave-uniform i64->f32 conversion that hipcc 7.2 cannot select on the SALU).
  return ldexpf(1.0f, bits) - 1.0f;
}

// ---- wave / block reductions (64-lane wavefront shuffles, LDS only across waves) -------------
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// block-wide min & max for a 256-thread block; result valid in every thread
__device__ __forceinline__ void block_minmax(float& mn, float& mx, float* lds /*[8]*/) {
  mn = wave_min(mn);
  mx = wave_max(mx);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) { lds[w] = mn; lds[4 + w] = mx; }
  __syncthreads();
  mn = fminf(fminf(lds[0], lds[1]), fminf(lds[2], lds[3]));
  mx = fmaxf(fmaxf(lds[4], lds[5]), fmaxf(lds[6], lds[7]));
  __syncthreads();
}
__device__ __forceinline__ float block_sum(float v, float* lds /*[4]*/) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) lds[w] = v;
  __syncthreads();
  v = (lds[0] + lds[1]) + (lds[2] + lds[3]);
  __syncthreads();
  return v;
}

static inline int pf_grid_for(int64_t n, int per_block) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > PF_MAX_GRID) g = PF_MAX_GRID;
  return (int)g;
}
