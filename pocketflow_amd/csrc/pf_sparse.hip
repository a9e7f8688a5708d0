// K7-K9: weight-sparsification mask refresh (backup merge, k-th |w| by radix select, mask/apply),
// sparsity counting, and the channel-pruning per-channel gradient masks.
//
// Reference semantics restated here (paths under /root/reference):
//   learners/weight_sparsification/learner.py:51-65, 283-288
//   learners/channel_pruning/learner.py:406-419
#include "pf_common.h"

// bkup <- where(mask > 0.5, var, bkup) ; abs_out <- |bkup|        (ws learner.py:283, 285)
__global__ __launch_bounds__(PF_THREADS) void k_ws_bkup_merge_abs(const float* __restrict__ var,
                                                                  float* __restrict__ bkup,
                                                                  const float* __restrict__ mask,
                                                                  float* __restrict__ abs_out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * PF_THREADS) {
    const float b = (mask[i] > 0.5f) ? var[i] : bkup[i];
    bkup[i] = b;
    abs_out[i] = fabsf(b);
  }
}
extern "C" int pf_ws_bkup_merge_abs(const float* var, float* bkup, const float* mask, float* abs_out,
                                    int64_t n, void* stream) {
  if (n <= 0) return 0;
  k_ws_bkup_merge_abs<<<pf_grid_for(n, PF_THREADS * 4), PF_THREADS, 0, (hipStream_t)stream>>>(var, bkup, mask, abs_out, n);
  PF_LAUNCH_CHECK();
  return 0;
}

// mask <- float(|bkup| > thr) ; var <- bkup * mask                 (ws learner.py:286-288)
__global__ __launch_bounds__(PF_THREADS) void k_ws_mask_apply(float* __restrict__ var,
                                                              const float* __restrict__ bkup,
                                                              float* __restrict__ mask,
                                                              const float* __restrict__ thr, int64_t n) {
  const float t = *thr;
  for (int64_t i = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * PF_THREADS) {
    const float b = bkup[i];
    const float m = (fabsf(b) > t) ? 1.0f : 0.0f;
    mask[i] = m;
    var[i] = b * m;
  }
}
extern "C" int pf_ws_mask_apply(float* var, const float* bkup, float* mask, const float* thr,
                                int64_t n, void* stream) {
  if (n <= 0) return 0;
  k_ws_mask_apply<<<pf_grid_for(n, PF_THREADS * 4), PF_THREADS, 0, (hipStream_t)stream>>>(var, bkup, mask, thr, n);
  PF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// k-th largest of a non-negative float array: 4-pass MSB radix select on the IEEE bit pattern
// (monotone for non-negative floats).  Equivalent to "sort descending, take [k]" of
// tf.contrib.distributions.percentile without sorting.  Exact, order-independent.
// workspace layout (uint32): [0..255] histogram, [256] prefix, [257] prefix mask, [258] k (lo),
// [259] k (hi).
// ---------------------------------------------------------------------------------------------
__global__ void k_kth_init(uint32_t* ws, uint64_t k) {
  for (int i = threadIdx.x; i < 256; i += blockDim.x) ws[i] = 0;
  if (threadIdx.x == 0) { ws[256] = 0; ws[257] = 0; ws[258] = (uint32_t)k; ws[259] = (uint32_t)(k >> 32); }
}
__global__ __launch_bounds__(PF_THREADS) void k_kth_hist(const float* __restrict__ a, int64_t n,
                                                         uint32_t* __restrict__ ws, int shift) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t prefix = ws[256], pmask = ws[257];
  for (int64_t i = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * PF_THREADS) {
    const uint32_t u = __float_as_uint(a[i]) & 0x7FFFFFFFu;
    if ((u & pmask) == prefix) atomicAdd(&h[(u >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&ws[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_kth_scan(uint32_t* ws, int shift, float* out, int last) {
  if (threadIdx.x == 0) {
    uint64_t k = (uint64_t)ws[258] | ((uint64_t)ws[259] << 32);
    uint64_t cum = 0;
    int digit = 0;
    for (int b = 255; b >= 0; --b) {          // descending order
      const uint64_t c = ws[b];
      if (k < cum + c) { digit = b; break; }
      cum += c;
    }
    k -= cum;
    const uint32_t prefix = ws[256] | ((uint32_t)digit << shift);
    ws[256] = prefix;
    ws[257] |= (0xFFu << shift);
    ws[258] = (uint32_t)k;
    ws[259] = (uint32_t)(k >> 32);
    if (last) *out = __uint_as_float(prefix);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) ws[i] = 0;
}
extern "C" int pf_kth_largest_nonneg(const float* a, int64_t n, int64_t k_desc_index, float* out,
                                     uint32_t* workspace, void* stream) {
  if (n <= 0 || k_desc_index < 0 || k_desc_index >= n) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  k_kth_init<<<1, 256, 0, st>>>(workspace, (uint64_t)k_desc_index);
  const int grid = pf_grid_for(n, PF_THREADS * 8);
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    k_kth_hist<<<grid, PF_THREADS, 0, st>>>(a, n, workspace, shift);
    k_kth_scan<<<1, 256, 0, st>>>(workspace, shift, out, pass == 3);
  }
  PF_LAUNCH_CHECK();
  return 0;
}

// count_nonzero (calc_prune_ratio, ws learner.py:51-65); *out must be zeroed by the caller
__global__ __launch_bounds__(PF_THREADS) void k_count_nonzero(const float* __restrict__ x, int64_t n,
                                                              unsigned long long* __restrict__ out) {
  __shared__ float lds[4];
  unsigned int c = 0;
  for (int64_t i = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * PF_THREADS)
    c += (x[i] != 0.0f) ? 1u : 0u;
  // per-thread counts < 2^24 are exact in float32 only up to 16M; reduce as integers instead
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  __shared__ unsigned int wc[4];
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(wc[0] + wc[1] + wc[2] + wc[3]));
  (void)lds;
}
extern "C" int pf_count_nonzero(const float* x, int64_t n, unsigned long long* out, void* stream) {
  if (n <= 0) return 0;
  k_count_nonzero<<<pf_grid_for(n, PF_THREADS * 8), PF_THREADS, 0, (hipStream_t)stream>>>(x, n, out);
  PF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Channel-pruning masks over a KRSC kernel [O][RS][I]: element kept iff keep_out[o] && keep_in[i].
// keep vectors are staged in LDS once per block.
// ---------------------------------------------------------------------------------------------
template <bool BUILD>
__global__ __launch_bounds__(PF_THREADS) void k_cp_mask(float* __restrict__ t,
                                                        const uint8_t* __restrict__ keep_in,
                                                        const uint8_t* __restrict__ keep_out, int O,
                                                        int RS, int I) {
  extern __shared__ uint8_t s_keep[];          // [I] then [O]
  uint8_t* s_in = s_keep;
  uint8_t* s_out = s_keep + I;
  for (int i = threadIdx.x; i < I; i += PF_THREADS) s_in[i] = keep_in[i];
  for (int o = threadIdx.x; o < O; o += PF_THREADS) s_out[o] = keep_out[o];
  __syncthreads();
  const int64_t n = (int64_t)O * RS * I;
  const int64_t L = (int64_t)RS * I;
  for (int64_t e = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * PF_THREADS) {
    const int o = (int)(e / L);
    const int i = (int)(e % I);
    const float m = (s_out[o] && s_in[i]) ? 1.0f : 0.0f;
    if (BUILD) t[e] = m;
    else t[e] = t[e] * m;
  }
}
extern "C" int pf_cp_build_mask(float* mask, const uint8_t* keep_in, const uint8_t* keep_out, int O,
                                int RS, int I, void* stream) {
  const int64_t n = (int64_t)O * RS * I;
  if (n <= 0) return 0;
  k_cp_mask<true><<<pf_grid_for(n, PF_THREADS * 4), PF_THREADS, (size_t)(I + O), (hipStream_t)stream>>>(mask, keep_in, keep_out, O, RS, I);
  PF_LAUNCH_CHECK();
  return 0;
}
extern "C" int pf_cp_mask_grad(float* g, const uint8_t* keep_in, const uint8_t* keep_out, int O,
                               int RS, int I, void* stream) {
  const int64_t n = (int64_t)O * RS * I;
  if (n <= 0) return 0;
  k_cp_mask<false><<<pf_grid_for(n, PF_THREADS * 4), PF_THREADS, (size_t)(I + O), (hipStream_t)stream>>>(g, keep_in, keep_out, O, RS, I);
  PF_LAUNCH_CHECK();
  return 0;
}
