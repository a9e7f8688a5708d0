// The stochastic proximal-gradient step of the 'chn-pruned-gpu' learner, fused: one gradient step on a convolution kernel followed by
// the group-lasso shrinkage of its INPUT channels,
//     W' = W - lr * G;   n_c = sqrt(sum_{o,r,s} W'[o][r][s][c]^2);   tau = percentile(n, p) (nearest rank);
//     W  <- W' * max(1 - tau / n_c, 0)
// (reference learners/channel_pruning_gpu/learner.py:379-383: var_prnd_new / var_norm / threshold / shrk_vec / prune_op; one TF op per
// rounding: the subtraction, the square, the division, the subtraction from one, the maximum and the product are each rounded once).
// HBM-bound.  KRSC storage makes the kernel a [rows = O*R*S][I] row-major matrix with the group index c innermost, so the channel
// norms are COLUMN sums: pass 1 reads W and G once (rows split over G workgroups, per-thread column accumulators, float32 partial
// sums [G][I], fixed-order reduction: deterministic), the threshold comes from pf_kth_largest_nonneg over the I norms, pass 2 reads W
// and G again and writes W.  12 + 12 bytes per parameter for float32 gradients; round 3 ran five torch ops (8 passes) here.
#include "pf_common.h"

#define PX_T 256
#define PX_MAXG 256

// partial[g][c] = sum over the workgroup's rows of (w - lr * grad)^2
template <typename TG>
__global__ __launch_bounds__(PX_T) void k_prox_sumsq(const float* __restrict__ w, const TG* __restrict__ g, float lr, int64_t rows, int I,
                                                     float* __restrict__ partial) {
  const int G = gridDim.y;
  const int c = blockIdx.x * PX_T + threadIdx.x;
  if (c >= I) return;
  const int64_t per = (rows + G - 1) / G;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    const float v = w[r * I + c] - lr * load_one<TG>(g + r * I + c);
    s = s + v * v;
  }
  partial[(int64_t)blockIdx.y * I + c] = s;
}

__global__ __launch_bounds__(PX_T) void k_prox_norm(const float* __restrict__ partial, int G, int I, float* __restrict__ norms) {
  const int c = blockIdx.x * PX_T + threadIdx.x;
  if (c >= I) return;
  float s = partial[c];
  for (int g = 1; g < G; ++g) s = s + partial[(int64_t)g * I + c];          // ascending: deterministic
  norms[c] = sqrtf(s);
}

// w <- (w - lr * grad) * max(1 - thr / norm_c, 0);  0 / 0 (a dead channel under a zero threshold) -> 0, as tf.maximum(nan, 0) is not
// what the reference can reach either: its percentile of an all-zero norm vector leaves 0 * nan = nan, which the learner never
// produces because a dead channel stays dead (0 * anything finite); the guard below makes that explicit
template <typename TG>
__global__ __launch_bounds__(PX_T) void k_prox_apply(float* __restrict__ w, const TG* __restrict__ g, float lr, int64_t n, int I,
                                                     const float* __restrict__ norms, const float* __restrict__ thr) {
  const float t = thr[0];
  for (int64_t i = (int64_t)blockIdx.x * PX_T + threadIdx.x; i < n; i += (int64_t)gridDim.x * PX_T) {
    const int c = (int)(i % I);
    const float nc = norms[c];
    float shrk = fmaxf(1.0f - t / nc, 0.0f);
    if (!(shrk == shrk)) shrk = 0.0f;
    const float v = w[i] - lr * load_one<TG>(g + i);
    w[i] = v * shrk;
  }
}

extern "C" int pf_prox_groups(int64_t rows, int I) {
  int64_t g = (4096 + I - 1) / I;                            // ~4096 column-lanes in flight over the chip ...
  if (g > rows / 8) g = rows / 8;                            // ... and at least 8 rows per workgroup
  if (g < 1) g = 1;
  if (g > PX_MAXG) g = PX_MAXG;
  return (int)g;
}

// norms[I] of W' = W - lr * G.  partial: float32 workspace of pf_prox_groups(rows, I) * I elements.
extern "C" int pf_prox_norms(const float* w, const void* g, int g_dtype, float lr, int64_t rows, int I, float* partial, float* norms,
                             void* stream) {
  if (rows <= 0 || I <= 0 || w == nullptr || g == nullptr || partial == nullptr || norms == nullptr) return (int)hipErrorInvalidValue;
  const int G = pf_prox_groups(rows, I);
  const dim3 grid((unsigned)((I + PX_T - 1) / PX_T), (unsigned)G, 1);
  hipStream_t st = (hipStream_t)stream;
  if (g_dtype == PF_F32) k_prox_sumsq<float><<<grid, PX_T, 0, st>>>(w, (const float*)g, lr, rows, I, partial);
  else if (g_dtype == PF_BF16) k_prox_sumsq<bf16_t><<<grid, PX_T, 0, st>>>(w, (const bf16_t*)g, lr, rows, I, partial);
  else return (int)hipErrorInvalidValue;
  PF_LAUNCH_CHECK();
  k_prox_norm<<<(I + PX_T - 1) / PX_T, PX_T, 0, st>>>(partial, G, I, norms);
  PF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pf_prox_apply(float* w, const void* g, int g_dtype, float lr, int64_t rows, int I, const float* norms, const float* thr,
                             void* stream) {
  if (rows <= 0 || I <= 0 || w == nullptr || g == nullptr || norms == nullptr || thr == nullptr) return (int)hipErrorInvalidValue;
  const int64_t n = rows * I;
  const int grid = pf_grid_for(n, PX_T);
  hipStream_t st = (hipStream_t)stream;
  if (g_dtype == PF_F32) k_prox_apply<float><<<grid, PX_T, 0, st>>>(w, (const float*)g, lr, n, I, norms, thr);
  else if (g_dtype == PF_BF16) k_prox_apply<bf16_t><<<grid, PX_T, 0, st>>>(w, (const bf16_t*)g, lr, n, I, norms, thr);
  else return (int)hipErrorInvalidValue;
  PF_LAUNCH_CHECK();
  return 0;
}
