// K13 fused with K4: batch-norm (training / inference) + ReLU/ReLU6 + activation fake-quant over
// NHWC activations viewed as a [rows][C] matrix, forward and backward.
//
//   forward : pass 1 reads x once -> per-channel {sum, sumsq, min, max} (block partials)
//             finalize            -> mean/var, scale/shift, moving stats, and the EXACT whole-tensor
//                                    min/max of y = act(scale*x+shift) from per-channel min/max of x
//                                    (y is monotone in x per channel), written to the min/max slot
//             pass 2 reads x, writes q = fake_quant(y)
//   i.e. 2 reads + 1 write per element instead of the reference's ~9 separate TF elementwise /
//   reduction kernels (batch_norm, relu, reduce_max, reduce_min, sub, div, mul, round, div, mul, add).
//   backward: pass 1 reads dq, x -> per-channel {sum dy, sum dy*xhat}; pass 2 reads dq, x, writes dx.
//             The ReLU mask is recomputed from x, so neither y nor a mask tensor is ever stored.
//
// Reference semantics restated here (paths under /root/reference; BN maths is TF's fused kernel):
//   utils/external/resnet_model.py:55-62   tf.layers.batch_normalization(momentum=.997, eps=1e-5, fused)
//   utils/external/mobilenet_v1.py:428-477 slim.batch_norm(decay=.9997, epsilon=1e-3)
//   learners/uniform_quantization/utils.py:51-79  activation fake-quant after every Relu / Relu6
#include "pf_common.h"

// Fast path: C % 8 == 0, G = C/8 <= 256, 256 % G == 0 (C = 8 .. 2048, powers of two): a thread owns
// one group of 8 consecutive channels and walks rows; a wavefront reads 1 KiB contiguous.
static inline bool bn_fast_ok(int C) {
  if (C % 8) return false;
  const int G = C / 8;
  if (G < 8) return (8 % G) == 0;                        // C = 8, 16, 32: a single slab
  return (C % 64) == 0 && G <= 256 && (256 % G) == 0;    // streaming kernels keep the (256 % G) map
}

// ---------------------------------------------------------------------------------------------
// forward pass 1
// ---------------------------------------------------------------------------------------------
// 2-D decomposition: a 1024-thread workgroup (16 wavefronts) owns a SLAB of up to 64 channels
// (CG = min(C/8, 8) groups of 8 channels, 16 B per lane) and one of `nsplit` interleaved row ranges;
// 1024/CG row-lanes walk the rows.  Per-workgroup partials are only 4 x 64 floats, so the partial
// buffer [nsplit][4][C] stays ~1000x smaller than the tensor and the finalize pass is a few us.
#define BN_BIG 1024
__device__ __forceinline__ int bn_cg_of(int C) { return (C >> 3) < 8 ? (C >> 3) : 8; }

// KIND per statistic: 0 = sum, 1 = min, 2 = max
template <int KIND> __device__ __forceinline__ float bn_comb(float a, float b) {
  return KIND == 0 ? (a + b) : (KIND == 1 ? fminf(a, b) : fmaxf(a, b));
}
template <int KIND>
__device__ __forceinline__ void bn_wave_reduce8(float (&a)[8], int CG) {
  // lanes with equal (lane % CG) hold the same channel group: butterfly over the other lane bits
  for (int m = CG; m < 64; m <<= 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = bn_comb<KIND>(a[j], __shfl_xor(a[j], m, 64));
  }
}
template <int KIND>
__device__ __forceinline__ void bn_lds_put(const float (&a)[8], int st, int NS, int CG, float* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < CG) {
#pragma unroll
    for (int j = 0; j < 8; ++j) lds[(wave * NS + st) * 64 + lane * 8 + j] = a[j];
  }
}
template <int KIND>
__device__ __forceinline__ void bn_lds_finish(int st, int NS, int slabC, const float* lds, float* out) {
  // threads [st*slabC, (st+1)*slabC) combine the 16 per-wave partials of statistic `st`
  const int t = (int)threadIdx.x - st * slabC;
  if (t >= 0 && t < slabC) {
    float a = lds[st * 64 + t];
    for (int w = 1; w < BN_BIG / 64; ++w) a = bn_comb<KIND>(a, lds[(w * NS + st) * 64 + t]);
    out[t] = a;
  }
}

template <typename T>
__global__ __launch_bounds__(BN_BIG) void k_bn_stats_fast(const T* __restrict__ x, int64_t rows, int C,
                                                          float* __restrict__ partial, int nslab, int nsplit) {
  __shared__ float lds[(BN_BIG / 64) * 4 * 64];
  const int CG = bn_cg_of(C), slabC = CG * 8, RL = BN_BIG / CG;
  const int slab = blockIdx.x % nslab, split = blockIdx.x / nslab;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  const int c0 = slab * slabC + cg * 8;
  float piv[8];
  load8<T>(x + c0, piv);                          // pivot = row 0 (shifted sums: no cancellation)
  float s[8], ss[8], mn[8], mx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; ss[j] = 0.f; mn[j] = INFINITY; mx[j] = -INFINITY; }
#pragma unroll 2
  for (int64_t r = (int64_t)split * RL + rl; r < rows; r += (int64_t)nsplit * RL) {
    float v[8];
    load8<T>(x + r * C + c0, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = v[j] - piv[j];
      s[j] += d;
      ss[j] = fmaf(d, d, ss[j]);
      mn[j] = pf_acc_min(mn[j], v[j]);
      mx[j] = pf_acc_max(mx[j], v[j]);
    }
  }
  bn_wave_reduce8<0>(s, CG); bn_wave_reduce8<0>(ss, CG); bn_wave_reduce8<1>(mn, CG); bn_wave_reduce8<2>(mx, CG);
  bn_lds_put<0>(s, 0, 4, CG, lds); bn_lds_put<0>(ss, 1, 4, CG, lds);
  bn_lds_put<1>(mn, 2, 4, CG, lds); bn_lds_put<2>(mx, 3, 4, CG, lds);
  __syncthreads();
  float* out = partial + (int64_t)split * 4 * C + slab * slabC;
  bn_lds_finish<0>(0, 4, slabC, lds, out);
  bn_lds_finish<0>(1, 4, slabC, lds, out + C);
  bn_lds_finish<1>(2, 4, slabC, lds, out + 2 * C);
  bn_lds_finish<2>(3, 4, slabC, lds, out + 3 * C);
}

template <typename T>
__global__ __launch_bounds__(PF_THREADS) void k_bn_stats_generic(const T* __restrict__ x, int64_t rows, int C,
                                                                 float* __restrict__ partial) {
  // thread = channel (grid.x over channel tiles), grid.y = row split
  const int c = blockIdx.x * PF_THREADS + threadIdx.x;
  if (c >= C) return;
  const float piv = load_one<T>(x + c);
  float s = 0.f, ss = 0.f, mn = INFINITY, mx = -INFINITY;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const float v = load_one<T>(x + r * C + c);
    const float d = v - piv;
    s += d; ss = fmaf(d, d, ss);
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
  float* out = partial + (int64_t)blockIdx.y * 4 * C;
  out[c] = s; out[C + c] = ss; out[2 * C + c] = mn; out[3 * C + c] = mx;
}

extern "C" int pf_bn_stats(const void* x, int dtype, int64_t rows, int C, float* partial,
                           int n_blocks, void* stream) {
  if (rows <= 0 || C <= 0 || n_blocks <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  if (bn_fast_ok(C) && pf_aligned16(x)) {
    const int CG = (C / 8) < 8 ? (C / 8) : 8;
    const int nslab = C / (CG * 8);
    if (dtype == PF_F32) k_bn_stats_fast<float><<<nslab * n_blocks, BN_BIG, 0, st>>>((const float*)x, rows, C, partial, nslab, n_blocks);
    else if (dtype == PF_BF16) k_bn_stats_fast<bf16_t><<<nslab * n_blocks, BN_BIG, 0, st>>>((const bf16_t*)x, rows, C, partial, nslab, n_blocks);
    else return (int)hipErrorInvalidValue;
  } else {
    dim3 grid((C + PF_THREADS - 1) / PF_THREADS, n_blocks);
    if (dtype == PF_F32) k_bn_stats_generic<float><<<grid, PF_THREADS, 0, st>>>((const float*)x, rows, C, partial);
    else if (dtype == PF_BF16) k_bn_stats_generic<bf16_t><<<grid, PF_THREADS, 0, st>>>((const bf16_t*)x, rows, C, partial);
    else return (int)hipErrorInvalidValue;
  }
  PF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// finalize: 16 channels x 16 row-partial lanes per workgroup.  The kernel is a dependent-latency chain between a producer's
// statistics and the next launch that reads scale / shift (49 launches a step, nothing else of the student's stream in flight),
// so the levers are workgroups (C / 16 instead of C / 64) and loads in flight: a lane owns partial blocks part, part+16, ...
// and requests eight of them (32 loads) before it adds -- in ASCENDING block order, then the 16 lanes of a channel combine in
// ascending lane order through LDS: the summation order of rounds 1-5 (64 channels x 16 lanes, one block per trip), bit for bit.
// (A 64-lane tree per channel was 0.2 us faster and equally accurate -- every call within 1.5e-6 of a float64 two-pass result,
// tools/gpu/bn_finalize_audit.py -- but moved the bf16 gradient checks of the two small networks, whose per-variable bars
// compare noise with noise at the 0.05 level, across their margins: profiles/r06_bn_finalize_order.txt.)
// 13.8 -> 12.0 us per launch inside replays, 6.5 us launch by launch (profiles/r06_step_kernels_b256*.csv).
// ---------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ void y_range(float scale, float shift, float xmin, float xmax, float& ymin,
                                        float& ymax) {
  const float a = fmaf(scale, xmin, shift), b = fmaf(scale, xmax, shift);
  ymin = apply_act<ACT>(fminf(a, b));
  ymax = apply_act<ACT>(fmaxf(a, b));
}

#define BN_FIN_C 16                      // channels per block
#define BN_FIN_P 16                      // row-partial lanes per channel
#define BN_FIN_T (BN_FIN_P * BN_FIN_C)   // 256 threads
#define BN_FIN_U 8                       // partial blocks requested per trip
__global__ __launch_bounds__(BN_FIN_T) void k_bn_finalize(
    const float* __restrict__ partial, int n_blocks, int64_t rows, int C, const void* __restrict__ x_row0,
    int dtype, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ moving_mean, float* __restrict__ moving_var, float momentum, float eps,
    int training, int act, float* __restrict__ scale_shift, float* __restrict__ mean_invstd,
    uint32_t* __restrict__ slot) {
  __shared__ float l_s[BN_FIN_P][BN_FIN_C], l_ss[BN_FIN_P][BN_FIN_C], l_mn[BN_FIN_P][BN_FIN_C], l_mx[BN_FIN_P][BN_FIN_C];
  const int cl = threadIdx.x & (BN_FIN_C - 1), part = threadIdx.x / BN_FIN_C;
  const int c = blockIdx.x * BN_FIN_C + cl;
  const bool live = c < C;
  float s = 0.f, ss = 0.f, mn = INFINITY, mx = -INFINITY;
  // constants of the tail, requested before the partial sums so that their latency hides under the loop
  float piv = 0.f, g = 0.f, be = 0.f, mm = 0.f, mv = 0.f;
  if (live && part == 0) {
    g = gamma[c]; be = beta[c]; mm = moving_mean[c]; mv = moving_var[c];
    if (training) piv = (dtype == PF_F32) ? ((const float*)x_row0)[c] : bf16_to_f32(((const bf16_t*)x_row0)[c]);
  }
  if (live) {
    const int64_t step = (int64_t)BN_FIN_P * 4 * C;
    int b = part;
    for (; b + (BN_FIN_U - 1) * BN_FIN_P < n_blocks; b += BN_FIN_U * BN_FIN_P) {
      const float* p = partial + (int64_t)b * 4 * C + c;
      float v0[BN_FIN_U], v1[BN_FIN_U], v2[BN_FIN_U], v3[BN_FIN_U];
#pragma unroll
      for (int u = 0; u < BN_FIN_U; ++u) { v0[u] = p[u * step]; v1[u] = p[u * step + C]; v2[u] = p[u * step + 2 * C]; v3[u] = p[u * step + 3 * C]; }
#pragma unroll
      for (int u = 0; u < BN_FIN_U; ++u) {                 // ascending block order
        s += v0[u]; ss += v1[u];
        mn = fminf(mn, v2[u]); mx = fmaxf(mx, v3[u]);
      }
    }
    for (; b < n_blocks; b += BN_FIN_P) {
      const float* p = partial + (int64_t)b * 4 * C + c;
      s += p[0]; ss += p[C];
      mn = fminf(mn, p[2 * C]); mx = fmaxf(mx, p[3 * C]);
    }
  }
  l_s[part][cl] = s; l_ss[part][cl] = ss; l_mn[part][cl] = mn; l_mx[part][cl] = mx;
  __syncthreads();
  float ymin = INFINITY, ymax = -INFINITY;
  if (part == 0 && live) {
    s = l_s[0][cl]; ss = l_ss[0][cl]; mn = l_mn[0][cl]; mx = l_mx[0][cl];
#pragma unroll
    for (int p = 1; p < 16; ++p) {                       // fixed order: deterministic
      s += l_s[p][cl]; ss += l_ss[p][cl];
      mn = fminf(mn, l_mn[p][cl]); mx = fmaxf(mx, l_mx[p][cl]);
    }
    float mean, var;
    if (training) {
      const double n = (double)rows;
      const double m1 = (double)s / n;
      double v = (double)ss / n - m1 * m1;             // biased variance of (x - pivot) == of x
      if (v < 0.0) v = 0.0;
      mean = (float)((double)piv + m1);
      var = (float)v;
      // moving averages: tf.layers BN  moving -= (moving - batch) * (1 - momentum); the fused
      // kernel hands the UNBIASED variance to the moving average
      const float unbiased = (float)(v * (n / (n > 1.0 ? n - 1.0 : 1.0)));
      const float omm = 1.0f - momentum;
      moving_mean[c] = mm - (mm - mean) * omm;
      moving_var[c] = mv - (mv - unbiased) * omm;
    } else {
      mean = mm;
      var = mv;
    }
    const float invstd = 1.0f / sqrtf(var + eps);
    const float sc = g * invstd;
    const float sh = fmaf(-mean, sc, be);
    scale_shift[c] = sc;
    scale_shift[C + c] = sh;
    mean_invstd[c] = mean;
    mean_invstd[C + c] = invstd;
    if (act == PF_ACT_RELU) y_range<PF_ACT_RELU>(sc, sh, mn, mx, ymin, ymax);
    else if (act == PF_ACT_RELU6) y_range<PF_ACT_RELU6>(sc, sh, mn, mx, ymin, ymax);
    else y_range<PF_ACT_NONE>(sc, sh, mn, mx, ymin, ymax);
  }
  if (threadIdx.x < 64 && slot != nullptr) {             // wave 0 entire (lanes 16..63 carry the neutral elements)
    ymin = wave_min(ymin);
    ymax = wave_max(ymax);
    if (threadIdx.x == 0 && ymin <= ymax) {
      atomicMin(&slot[0], enc_f32(ymin));
      atomicMin(&slot[1], ~enc_f32(ymax));
    }
  }
}

extern "C" int pf_bn_finalize(const float* partial, int n_blocks, int64_t rows, int C,
                              const void* x_row0, int dtype, const float* gamma, const float* beta,
                              float* moving_mean, float* moving_var, float momentum, float eps,
                              int training, int act, float* scale_shift, float* mean_invstd,
                              uint32_t* slot, void* stream) {
  if (C <= 0) return (int)hipErrorInvalidValue;
  k_bn_finalize<<<(C + BN_FIN_C - 1) / BN_FIN_C, BN_FIN_T, 0, (hipStream_t)stream>>>(
      partial, n_blocks, rows, C, x_row0, dtype, gamma, beta, moving_mean, moving_var, momentum, eps,
      training, act, scale_shift, mean_invstd, slot);
  PF_LAUNCH_CHECK();
  return 0;
}

__global__ void k_bn_eval_scale_shift(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ mm, const float* __restrict__ mv, float eps,
                                      int C, float* __restrict__ ss) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float invstd = 1.0f / sqrtf(mv[c] + eps);
    const float sc = gamma[c] * invstd;
    ss[c] = sc;
    ss[C + c] = fmaf(-mm[c], sc, beta[c]);
  }
}
extern "C" int pf_bn_eval_scale_shift(const float* gamma, const float* beta, const float* moving_mean,
                                      const float* moving_var, float eps, int C, float* scale_shift,
                                      void* stream) {
  k_bn_eval_scale_shift<<<(C + 255) / 256, 256, 0, (hipStream_t)stream>>>(gamma, beta, moving_mean, moving_var, eps, C, scale_shift);
  PF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// forward pass 2:  q = fake_quant(act(scale*x + shift))
// ---------------------------------------------------------------------------------------------
template <typename T, int ACT, bool FAST>
__global__ __launch_bounds__(PF_THREADS) void k_bn_apply(const T* __restrict__ x, T* __restrict__ q,
                                                         int64_t rows, int C,
                                                         const float* __restrict__ scale_shift,
                                                         const uint32_t* __restrict__ slot, float k,
                                                         int quantize) {
  float alpha = 1.f, beta = 0.f;
  if (quantize) slot_alpha_beta(slot, alpha, beta);
  if (FAST) {
    const int G = C >> 3, RPS = PF_THREADS / G;
    const int cg = threadIdx.x % G, rsub = threadIdx.x / G;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale_shift[(cg << 3) + j]; sh[j] = scale_shift[C + (cg << 3) + j]; }
#pragma unroll 2
    for (int64_t r = (int64_t)blockIdx.x * RPS + rsub; r < rows; r += (int64_t)gridDim.x * RPS) {
      float v[8];
      const int64_t off = r * C + (cg << 3);
      load8<T>(x + off, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float y = apply_act<ACT>(fmaf(sc[j], v[j], sh[j]));
        v[j] = quantize ? uq_point(y, alpha, beta, k) : y;
      }
      store8<T>(q + off, v);
    }
  } else {
    const int64_t n = rows * C;
    for (int64_t e = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * PF_THREADS) {
      const int c = (int)(e % C);
      const float y = apply_act<ACT>(fmaf(scale_shift[c], load_one<T>(x + e), scale_shift[C + c]));
      store_one<T>(q + e, quantize ? uq_point(y, alpha, beta, k) : y);
    }
  }
}

template <typename T>
static int launch_bn_apply(const T* x, T* q, int64_t rows, int C, const float* ss, int act,
                           const uint32_t* slot, float k, int quantize, hipStream_t st) {
  const bool fast = bn_fast_ok(C) && pf_aligned16(x) && pf_aligned16(q);
  const int grid = fast ? pf_grid_for(rows, (PF_THREADS / (C / 8)) * 2) : pf_grid_for(rows * C, PF_THREADS * 4);
#define PF_BA(ACTV)                                                                                     \
  do {                                                                                                  \
    if (fast) k_bn_apply<T, ACTV, true><<<grid, PF_THREADS, 0, st>>>(x, q, rows, C, ss, slot, k, quantize);   \
    else k_bn_apply<T, ACTV, false><<<grid, PF_THREADS, 0, st>>>(x, q, rows, C, ss, slot, k, quantize);       \
  } while (0)
  if (act == PF_ACT_RELU) PF_BA(PF_ACT_RELU);
  else if (act == PF_ACT_RELU6) PF_BA(PF_ACT_RELU6);
  else PF_BA(PF_ACT_NONE);
#undef PF_BA
  PF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pf_bn_act_quant_apply(const void* x, void* q, int dtype, int64_t rows, int C,
                                     const float* scale_shift, int act, const uint32_t* slot, int bits,
                                     int quantize, void* stream) {
  if (rows <= 0 || C <= 0) return (int)hipErrorInvalidValue;
  if (quantize && (bits < 1 || bits > 32 || slot == nullptr)) return (int)hipErrorInvalidValue;
  const float k = uq_k_of_bits(quantize ? bits : 8);
  if (dtype == PF_F32) return launch_bn_apply<float>((const float*)x, (float*)q, rows, C, scale_shift, act, slot, k, quantize, (hipStream_t)stream);
  if (dtype == PF_BF16) return launch_bn_apply<bf16_t>((const bf16_t*)x, (bf16_t*)q, rows, C, scale_shift, act, slot, k, quantize, (hipStream_t)stream);
  return (int)hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// backward pass 1: per-channel sum(dy), sum(dy * xhat),  dy = dq * act'(scale*x+shift)
// ---------------------------------------------------------------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(BN_BIG) void k_bn_bwd_stats_fast(const T* __restrict__ dq, const T* __restrict__ x,
                                                              int64_t rows, int C,
                                                              const float* __restrict__ scale_shift,
                                                              const float* __restrict__ mean_invstd,
                                                              float* __restrict__ partial, int nslab, int nsplit) {
  __shared__ float lds[(BN_BIG / 64) * 2 * 64];
  const int CG = bn_cg_of(C), slabC = CG * 8, RL = BN_BIG / CG;
  const int slab = blockIdx.x % nslab, split = blockIdx.x / nslab;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  const int c0 = slab * slabC + cg * 8;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    sc[j] = scale_shift[c]; sh[j] = scale_shift[C + c];
    mu[j] = mean_invstd[c]; is[j] = mean_invstd[C + c];
    s1[j] = 0.f; s2[j] = 0.f;
  }
#pragma unroll 2
  for (int64_t r = (int64_t)split * RL + rl; r < rows; r += (int64_t)nsplit * RL) {
    float g[8], v[8];
    const int64_t off = r * C + c0;
    load8<T>(dq + off, g);
    load8<T>(x + off, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float dy = g[j] * act_mask<ACT>(fmaf(sc[j], v[j], sh[j]));
      s1[j] += dy;
      s2[j] = fmaf(dy, (v[j] - mu[j]) * is[j], s2[j]);
    }
  }
  bn_wave_reduce8<0>(s1, CG); bn_wave_reduce8<0>(s2, CG);
  bn_lds_put<0>(s1, 0, 2, CG, lds); bn_lds_put<0>(s2, 1, 2, CG, lds);
  __syncthreads();
  float* out = partial + (int64_t)split * 2 * C + slab * slabC;
  bn_lds_finish<0>(0, 2, slabC, lds, out);
  bn_lds_finish<0>(1, 2, slabC, lds, out + C);
}

template <typename T, int ACT>
__global__ __launch_bounds__(PF_THREADS) void k_bn_bwd_stats_generic(const T* __restrict__ dq, const T* __restrict__ x,
                                                                     int64_t rows, int C,
                                                                     const float* __restrict__ scale_shift,
                                                                     const float* __restrict__ mean_invstd,
                                                                     float* __restrict__ partial) {
  const int c = blockIdx.x * PF_THREADS + threadIdx.x;
  if (c >= C) return;
  const float sc = scale_shift[c], sh = scale_shift[C + c], mu = mean_invstd[c], is = mean_invstd[C + c];
  float s1 = 0.f, s2 = 0.f;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const float v = load_one<T>(x + r * C + c);
    const float dy = load_one<T>(dq + r * C + c) * act_mask<ACT>(fmaf(sc, v, sh));
    s1 += dy;
    s2 = fmaf(dy, (v - mu) * is, s2);
  }
  float* out = partial + (int64_t)blockIdx.y * 2 * C;
  out[c] = s1; out[C + c] = s2;
}

template <typename T>
static int launch_bn_bwd_stats(const T* dq, const T* x, int64_t rows, int C, const float* ss,
                               const float* mi, int act, float* partial, int n_blocks, hipStream_t st) {
  const bool fast = bn_fast_ok(C) && pf_aligned16(x) && pf_aligned16(dq);
  const int CG = (C / 8) < 8 ? (C / 8) : 8;
  const int nslab = fast ? C / (CG * 8) : 1;
  dim3 ggrid((C + PF_THREADS - 1) / PF_THREADS, n_blocks);
#define PF_BS(ACTV)                                                                                                       \
  do {                                                                                                                    \
    if (fast) k_bn_bwd_stats_fast<T, ACTV><<<nslab * n_blocks, BN_BIG, 0, st>>>(dq, x, rows, C, ss, mi, partial, nslab, n_blocks);  \
    else k_bn_bwd_stats_generic<T, ACTV><<<ggrid, PF_THREADS, 0, st>>>(dq, x, rows, C, ss, mi, partial);                  \
  } while (0)
  if (act == PF_ACT_RELU) PF_BS(PF_ACT_RELU);
  else if (act == PF_ACT_RELU6) PF_BS(PF_ACT_RELU6);
  else PF_BS(PF_ACT_NONE);
#undef PF_BS
  PF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pf_bn_bwd_stats(const void* dq, const void* x, int dtype, int64_t rows, int C,
                               const float* scale_shift, const float* mean_invstd, int act,
                               float* partial, int n_blocks, void* stream) {
  if (rows <= 0 || C <= 0 || n_blocks <= 0) return (int)hipErrorInvalidValue;
  if (dtype == PF_F32) return launch_bn_bwd_stats<float>((const float*)dq, (const float*)x, rows, C, scale_shift, mean_invstd, act, partial, n_blocks, (hipStream_t)stream);
  if (dtype == PF_BF16) return launch_bn_bwd_stats<bf16_t>((const bf16_t*)dq, (const bf16_t*)x, rows, C, scale_shift, mean_invstd, act, partial, n_blocks, (hipStream_t)stream);
  return (int)hipErrorInvalidValue;
}

__global__ __launch_bounds__(PF_THREADS) void k_bn_bwd_finalize(const float* __restrict__ partial, int n_blocks,
                                                                int C, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
  // 16 channels x 16 partial-row lanes per workgroup: the kernel is a dependent-latency chain (n_blocks <= 256 rows of
  // 2*C floats), so the lever is parallelism per channel -- 64 channels x 4 lanes took 13 us per launch, 49 launches a step.
  // Fixed summation order (lane-strided partial sums, then a fixed tree): deterministic.
  __shared__ float l1[16][17], l2[16][17];
  const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int b = part; b < n_blocks; b += 16) {
      const float* p = partial + (int64_t)b * 2 * C;
      s1 += p[c]; s2 += p[C + c];
    }
  l1[part][cl] = s1; l2[part][cl] = s2;
  __syncthreads();
  if (part == 0 && c < C) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a += l1[r][cl]; b += l2[r][cl]; }
    dbeta[c] = a;
    dgamma[c] = b;
  }
}

extern "C" int pf_bn_bwd_finalize(const float* partial, int n_blocks, int C, float* dgamma, float* dbeta,
                                  void* stream) {
  k_bn_bwd_finalize<<<(C + 15) / 16, PF_THREADS, 0, (hipStream_t)stream>>>(partial, n_blocks, C, dgamma, dbeta);
  PF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// backward pass 2: dx = gamma*invstd * (dy - dbeta/n - xhat * dgamma/n)
// ---------------------------------------------------------------------------------------------
template <typename T, int ACT, bool FAST>
__global__ __launch_bounds__(PF_THREADS) void k_bn_bwd_apply(const T* __restrict__ dq, const T* __restrict__ x,
                                                             T* __restrict__ dx, int64_t rows, int C,
                                                             const float* __restrict__ scale_shift,
                                                             const float* __restrict__ mean_invstd,
                                                             const float* __restrict__ dgamma,
                                                             const float* __restrict__ dbeta,
                                                             const T* __restrict__ addend) {
  // addend (optional): the gradient arriving through the block's identity shortcut, which shares x with
  // this BN -- dx_total = dx_bn + addend in the same pass (autograd would add them in a separate kernel)
  const float inv_n = 1.0f / (float)rows;
  if (FAST) {
    const int G = C >> 3, RPS = PF_THREADS / G;
    const int cg = threadIdx.x % G, rsub = threadIdx.x / G;
    float sc[8], sh[8], mu[8], is[8], a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = (cg << 3) + j;
      sc[j] = scale_shift[c]; sh[j] = scale_shift[C + c];
      mu[j] = mean_invstd[c]; is[j] = mean_invstd[C + c];
      a[j] = dbeta[c] * inv_n; b[j] = dgamma[c] * inv_n;
    }
#pragma unroll 2
    for (int64_t r = (int64_t)blockIdx.x * RPS + rsub; r < rows; r += (int64_t)gridDim.x * RPS) {
      float g[8], v[8];
      const int64_t off = r * C + (cg << 3);
      load8<T>(dq + off, g);
      load8<T>(x + off, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dy = g[j] * act_mask<ACT>(fmaf(sc[j], v[j], sh[j]));
        const float xh = (v[j] - mu[j]) * is[j];
        g[j] = sc[j] * (dy - a[j] - xh * b[j]);
      }
      if (addend != nullptr) {
        float ad[8];
        load8<T>(addend + off, ad);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] += ad[j];
      }
      store8<T>(dx + off, g);
    }
  } else {
    const int64_t n = rows * C;
    for (int64_t e = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * PF_THREADS) {
      const int c = (int)(e % C);
      const float v = load_one<T>(x + e);
      const float sc = scale_shift[c];
      const float dy = load_one<T>(dq + e) * act_mask<ACT>(fmaf(sc, v, scale_shift[C + c]));
      const float xh = (v - mean_invstd[c]) * mean_invstd[C + c];
      float o = sc * (dy - dbeta[c] * inv_n - xh * (dgamma[c] * inv_n));
      if (addend != nullptr) o += load_one<T>(addend + e);
      store_one<T>(dx + e, o);
    }
  }
}

template <typename T>
static int launch_bn_bwd_apply(const T* dq, const T* x, T* dx, int64_t rows, int C, const float* ss,
                               const float* mi, const float* dgamma, const float* dbeta, int act,
                               const T* addend, hipStream_t st) {
  const bool fast = bn_fast_ok(C) && pf_aligned16(x) && pf_aligned16(dq) && pf_aligned16(dx) && pf_aligned16(addend);
  const int grid = fast ? pf_grid_for(rows, (PF_THREADS / (C / 8)) * 2) : pf_grid_for(rows * C, PF_THREADS * 4);
#define PF_BB(ACTV)                                                                                                    \
  do {                                                                                                                 \
    if (fast) k_bn_bwd_apply<T, ACTV, true><<<grid, PF_THREADS, 0, st>>>(dq, x, dx, rows, C, ss, mi, dgamma, dbeta, addend);    \
    else k_bn_bwd_apply<T, ACTV, false><<<grid, PF_THREADS, 0, st>>>(dq, x, dx, rows, C, ss, mi, dgamma, dbeta, addend);        \
  } while (0)
  if (act == PF_ACT_RELU) PF_BB(PF_ACT_RELU);
  else if (act == PF_ACT_RELU6) PF_BB(PF_ACT_RELU6);
  else PF_BB(PF_ACT_NONE);
#undef PF_BB
  PF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pf_bn_bwd_apply_add(const void* dq, const void* x, const void* addend, void* dx, int dtype,
                                   int64_t rows, int C, const float* scale_shift, const float* mean_invstd,
                                   const float* dgamma, const float* dbeta, int act, void* stream) {
  if (rows <= 0 || C <= 0) return (int)hipErrorInvalidValue;
  if (dtype == PF_F32) return launch_bn_bwd_apply<float>((const float*)dq, (const float*)x, (float*)dx, rows, C, scale_shift, mean_invstd, dgamma, dbeta, act, (const float*)addend, (hipStream_t)stream);
  if (dtype == PF_BF16) return launch_bn_bwd_apply<bf16_t>((const bf16_t*)dq, (const bf16_t*)x, (bf16_t*)dx, rows, C, scale_shift, mean_invstd, dgamma, dbeta, act, (const bf16_t*)addend, (hipStream_t)stream);
  return (int)hipErrorInvalidValue;
}

extern "C" int pf_bn_bwd_apply(const void* dq, const void* x, void* dx, int dtype, int64_t rows, int C,
                               const float* scale_shift, const float* mean_invstd, const float* dgamma,
                               const float* dbeta, int act, void* stream) {
  return pf_bn_bwd_apply_add(dq, x, nullptr, dx, dtype, rows, C, scale_shift, mean_invstd, dgamma, dbeta, act, stream);
}
