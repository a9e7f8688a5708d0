// K1-K5: min/max calibration, uniform fake-quant (weights + activations), non-uniform (codebook)
// fake-quant.  HBM-bound integer/float elementwise + reduction work: 16-byte coalesced accesses,
// 64-lane shuffle reductions, LDS staging of per-channel (alpha, beta) / codebook vectors,
// order-independent atomicMin on encoded floats for grid-wide min/max.
//
// Reference semantics restated here (paths under /root/reference):
//   learners/uniform_quantization/utils.py:163-306   (__uniform_quantize, __scale, buckets)
//   learners/nonuniform_quantization/utils.py:168-347 (__nonuni_quantize, __bucket_quantize)
#include "pf_common.h"

// =============================================================================================
// K1  whole-tensor min/max of act(x)
// =============================================================================================
template <typename T, int ACT, bool VEC>
__global__ __launch_bounds__(PF_THREADS) void k_minmax_tensor(const T* __restrict__ x, int64_t n,
                                                              uint32_t* __restrict__ slot) {
  __shared__ float lds[8];
  float mn = INFINITY, mx = -INFINITY;
  const int64_t tid = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * PF_THREADS;
  int64_t done = 0;
  if (VEC) {
    const int64_t nv = n >> 3;
#pragma unroll 2
    for (int64_t i = tid; i < nv; i += stride) {
      float v[8];
      load8<T>(x + (i << 3), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = apply_act<ACT>(v[j]);
        mn = fminf(mn, t);
        mx = fmaxf(mx, t);
      }
    }
    done = nv << 3;
  }
  for (int64_t i = done + tid; i < n; i += stride) {
    float t = apply_act<ACT>(load_one<T>(x + i));
    mn = fminf(mn, t);
    mx = fmaxf(mx, t);
  }
  block_minmax(mn, mx, lds);
  if (threadIdx.x == 0 && mn <= mx) {
    atomicMin(&slot[0], enc_f32(mn));
    atomicMin(&slot[1], ~enc_f32(mx));
  }
}

template <typename T, bool VEC>
static void launch_minmax_tensor(const T* x, int64_t n, int act, uint32_t* slot, hipStream_t st) {
  const int grid = pf_grid_for(n, PF_THREADS * 16);
  switch (act) {
    case PF_ACT_RELU: k_minmax_tensor<T, PF_ACT_RELU, VEC><<<grid, PF_THREADS, 0, st>>>(x, n, slot); break;
    case PF_ACT_RELU6: k_minmax_tensor<T, PF_ACT_RELU6, VEC><<<grid, PF_THREADS, 0, st>>>(x, n, slot); break;
    default: k_minmax_tensor<T, PF_ACT_NONE, VEC><<<grid, PF_THREADS, 0, st>>>(x, n, slot); break;
  }
}

extern "C" int pf_minmax_slots_init(uint32_t* slots, int64_t n_pairs, void* stream) {
  if (n_pairs <= 0) return 0;
  return (int)hipMemsetAsync(slots, 0xFF, (size_t)n_pairs * 2 * sizeof(uint32_t), (hipStream_t)stream);
}

extern "C" int pf_minmax_tensor(const void* x, int64_t n, int dtype, int act, uint32_t* slot,
                                void* stream) {
  if (n <= 0) return 0;
  if (dtype != PF_F32 && dtype != PF_BF16) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = pf_aligned16(x);
  if (dtype == PF_F32) {
    if (vec) launch_minmax_tensor<float, true>((const float*)x, n, act, slot, st);
    else launch_minmax_tensor<float, false>((const float*)x, n, act, slot, st);
  } else {
    if (vec) launch_minmax_tensor<bf16_t, true>((const bf16_t*)x, n, act, slot, st);
    else launch_minmax_tensor<bf16_t, false>((const bf16_t*)x, n, act, slot, st);
  }
  PF_LAUNCH_CHECK();
  return 0;
}

__global__ void k_minmax_decode(const uint32_t* __restrict__ slots, int64_t n_pairs,
                                float* __restrict__ ab) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pairs) {
    float a, b;
    slot_alpha_beta(slots + 2 * i, a, b);
    ab[2 * i] = a;
    ab[2 * i + 1] = b;
  }
}
extern "C" int pf_minmax_decode(const uint32_t* slots, int64_t n_pairs, float* alpha_beta,
                                void* stream) {
  if (n_pairs <= 0) return 0;
  k_minmax_decode<<<(int)((n_pairs + 255) / 256), 256, 0, (hipStream_t)stream>>>(slots, n_pairs, alpha_beta);
  PF_LAUNCH_CHECK();
  return 0;
}

// =============================================================================================
// K2/K4  y = fake_quant(act(x)) with per-tensor (alpha, beta) from a slot
// =============================================================================================
template <typename TI, typename TO, int ACT, bool VEC>
__global__ __launch_bounds__(PF_THREADS) void k_uq_apply(const TI* __restrict__ x, TO* __restrict__ y,
                                                         int64_t n, const uint32_t* __restrict__ slot,
                                                         float k) {
  float alpha, beta;
  slot_alpha_beta(slot, alpha, beta);
  const int64_t tid = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * PF_THREADS;
  int64_t done = 0;
  if (VEC) {
    const int64_t nv = n >> 3;
#pragma unroll 2
    for (int64_t i = tid; i < nv; i += stride) {
      float v[8];
      load8<TI>(x + (i << 3), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = uq_point(apply_act<ACT>(v[j]), alpha, beta, k);
      store8<TO>(y + (i << 3), v);
    }
    done = nv << 3;
  }
  for (int64_t i = done + tid; i < n; i += stride)
    store_one<TO>(y + i, uq_point(apply_act<ACT>(load_one<TI>(x + i)), alpha, beta, k));
}

template <typename TI, typename TO, bool VEC>
static void launch_uq_apply(const TI* x, TO* y, int64_t n, int act, const uint32_t* slot, float k,
                            hipStream_t st) {
  const int grid = pf_grid_for(n, PF_THREADS * 16);
  switch (act) {
    case PF_ACT_RELU: k_uq_apply<TI, TO, PF_ACT_RELU, VEC><<<grid, PF_THREADS, 0, st>>>(x, y, n, slot, k); break;
    case PF_ACT_RELU6: k_uq_apply<TI, TO, PF_ACT_RELU6, VEC><<<grid, PF_THREADS, 0, st>>>(x, y, n, slot, k); break;
    default: k_uq_apply<TI, TO, PF_ACT_NONE, VEC><<<grid, PF_THREADS, 0, st>>>(x, y, n, slot, k); break;
  }
}

extern "C" int pf_uq_apply(const void* x, void* y, int64_t n, int in_dtype, int out_dtype, int act,
                           const uint32_t* slot, int bits, void* stream) {
  if (n <= 0) return 0;
  if (bits < 1 || bits > 32) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const float k = uq_k_of_bits(bits);
  const bool vec = pf_aligned16(x) && pf_aligned16(y);
#define PF_DISPATCH(TI, TO)                                                     \
  do {                                                                          \
    if (vec) launch_uq_apply<TI, TO, true>((const TI*)x, (TO*)y, n, act, slot, k, st);  \
    else launch_uq_apply<TI, TO, false>((const TI*)x, (TO*)y, n, act, slot, k, st);     \
  } while (0)
  if (in_dtype == PF_F32 && out_dtype == PF_F32) PF_DISPATCH(float, float);
  else if (in_dtype == PF_F32 && out_dtype == PF_BF16) PF_DISPATCH(float, bf16_t);
  else if (in_dtype == PF_BF16 && out_dtype == PF_BF16) PF_DISPATCH(bf16_t, bf16_t);
  else if (in_dtype == PF_BF16 && out_dtype == PF_F32) PF_DISPATCH(bf16_t, float);
  else return (int)hipErrorInvalidValue;
#undef PF_DISPATCH
  PF_LAUNCH_CHECK();
  return 0;
}

// backward: dx = g * act'(u)   (STE through the quantiser)
template <typename T, int ACT, bool VEC>
__global__ __launch_bounds__(PF_THREADS) void k_act_grad(const T* __restrict__ g, const T* __restrict__ u,
                                                         T* __restrict__ dx, int64_t n) {
  const int64_t tid = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * PF_THREADS;
  int64_t done = 0;
  if (VEC) {
    const int64_t nv = n >> 3;
#pragma unroll 2
    for (int64_t i = tid; i < nv; i += stride) {
      float a[8], b[8];
      load8<T>(g + (i << 3), a);
      load8<T>(u + (i << 3), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = a[j] * act_mask<ACT>(b[j]);
      store8<T>(dx + (i << 3), a);
    }
    done = nv << 3;
  }
  for (int64_t i = done + tid; i < n; i += stride)
    store_one<T>(dx + i, load_one<T>(g + i) * act_mask<ACT>(load_one<T>(u + i)));
}

template <typename T>
static int launch_act_grad(const T* g, const T* u, T* dx, int64_t n, int act, hipStream_t st) {
  const int grid = pf_grid_for(n, PF_THREADS * 16);
  const bool vec = pf_aligned16(g) && pf_aligned16(u) && pf_aligned16(dx);
#define PF_AG(ACTV)                                                                          \
  do {                                                                                       \
    if (vec) k_act_grad<T, ACTV, true><<<grid, PF_THREADS, 0, st>>>(g, u, dx, n);            \
    else k_act_grad<T, ACTV, false><<<grid, PF_THREADS, 0, st>>>(g, u, dx, n);               \
  } while (0)
  if (act == PF_ACT_RELU) PF_AG(PF_ACT_RELU);
  else if (act == PF_ACT_RELU6) PF_AG(PF_ACT_RELU6);
  else PF_AG(PF_ACT_NONE);
#undef PF_AG
  PF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pf_act_grad(const void* g, const void* u, void* dx, int64_t n, int dtype, int act,
                           void* stream) {
  if (n <= 0) return 0;
  if (dtype == PF_F32) return launch_act_grad<float>((const float*)g, (const float*)u, (float*)dx, n, act, (hipStream_t)stream);
  if (dtype == PF_BF16) return launch_act_grad<bf16_t>((const bf16_t*)g, (const bf16_t*)u, (bf16_t*)dx, n, act, (hipStream_t)stream);
  return (int)hipErrorInvalidValue;
}

// =============================================================================================
// Segment kernels: every weight tensor of the model in one launch.
// =============================================================================================

// storage index e -> bucket id
//   channel : bucket = output channel (KRSC: e / (RS*I))
//   split   : reference HWIO flat index f, bucket = f mod m  (uq utils.py:247-275: X[i][j] = flat[i*m+j])
// (32-bit index math: a single weight tensor has < 2^31 elements; 64-bit divisions are slow and
//  trip an instruction-selection bug in hipcc 7.2 when they sit under wave-uniform branches)
__device__ __forceinline__ uint32_t hwio_flat_index(const PfSeg& sg, uint32_t e) {
  if (sg.layout == 0) {                       // KRSC: e = (o*RS + rs)*I + i  ->  f = (rs*I + i)*O + o
    const uint32_t L = (uint32_t)sg.RS * (uint32_t)sg.I;
    const uint32_t o = e / L;
    const uint32_t rem = e - o * L;
    return rem * (uint32_t)sg.O + o;
  }
  // CRS (depthwise, O == 1): e = i*RS + rs  ->  f = rs*I + i
  const uint32_t i = e / (uint32_t)sg.RS;
  const uint32_t rs = e - i * (uint32_t)sg.RS;
  return rs * (uint32_t)sg.I + i;
}

__global__ __launch_bounds__(PF_THREADS) void k_seg_minmax(const float* __restrict__ w,
                                                           const PfSeg* __restrict__ segs,
                                                           const PfBlock* __restrict__ blocks,
                                                           uint32_t* __restrict__ slots) {
  __shared__ float lds[8];
  const PfBlock b = blocks[blockIdx.x];
  const PfSeg sg = segs[b.seg];
  const float* __restrict__ base = w + sg.offset;
  uint32_t* __restrict__ sl = slots + 2 * sg.slot_offset;

  if (sg.mode == PF_BUCKET_CHANNEL) {
    // one wavefront per output channel: a contiguous KRSC row of L = RS*I floats; no atomics
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = b.chunk * 4 + wave;
    if (row >= sg.O) return;
    const int64_t L = (int64_t)sg.RS * sg.I;
    const float* __restrict__ r = base + (int64_t)row * L;
    float mn = INFINITY, mx = -INFINITY;
    if ((L & 3) == 0) {
      for (int64_t i = lane * 4; i < L; i += 256) {
        float4 v = *reinterpret_cast<const float4*>(r + i);
        mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
        mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
      }
    } else {
      for (int64_t i = lane; i < L; i += 64) { float v = r[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane == 0) { sl[2 * row] = enc_f32(mn); sl[2 * row + 1] = ~enc_f32(mx); }
    return;
  }

  const int64_t e0 = (int64_t)b.chunk * PF_CHUNK;
  if (sg.mode == PF_BUCKET_SPLIT) {
    const uint32_t m = (uint32_t)sg.n_bucket;
    // a 4096-element chunk visits every bucket chunk/m times: with m <= 1024 the block reduces into LDS bins first
    // (order-independent integer atomicMin on the encoded values, so the result stays bit-deterministic) and issues at
    // most 2*m global atomics instead of 2 per element; larger m (up to ~9 k buckets per tensor) spreads the global
    // atomics thinly enough already
    __shared__ uint32_t bins[2 * 1024];
    const bool use_bins = m <= 1024u;
    if (use_bins) {
      for (uint32_t j = threadIdx.x; j < 2 * m; j += PF_THREADS) bins[j] = 0xFFFFFFFFu;
      __syncthreads();
    }
    for (int t = threadIdx.x; t < PF_CHUNK; t += PF_THREADS) {
      const int64_t e = e0 + t;
      if (e < sg.len) {
        const float v = base[e];
        const uint32_t j = hwio_flat_index(sg, (uint32_t)e) % m;
        if (use_bins) {
          atomicMin(&bins[2 * j], enc_f32(v));
          atomicMin(&bins[2 * j + 1], ~enc_f32(v));
        } else {
          atomicMin(&sl[2 * j], enc_f32(v));
          atomicMin(&sl[2 * j + 1], ~enc_f32(v));
        }
      }
    }
    if (use_bins) {
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < 2 * m; j += PF_THREADS)
        if (bins[j] != 0xFFFFFFFFu) atomicMin(&sl[j], bins[j]);
    }
    if (b.chunk == 0) {   // tail padding: copies of the LAST element join buckets f mod m, f >= len
      const float last = base[sg.len - 1];
      const uint32_t total = m * (uint32_t)sg.bucket_size;
      for (uint32_t f = (uint32_t)sg.len + threadIdx.x; f < total; f += PF_THREADS) {
        const uint32_t j = f % m;
        atomicMin(&sl[2 * j], enc_f32(last));
        atomicMin(&sl[2 * j + 1], ~enc_f32(last));
      }
    }
    return;
  }

  // per-tensor: 4096-element chunk, 4 x float4 per lane, block reduce, 2 atomics per block
  float mn = INFINITY, mx = -INFINITY;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int64_t e = e0 + it * 1024 + threadIdx.x * 4;
    if (e + 3 < sg.len) {
      float4 v = *reinterpret_cast<const float4*>(base + e);
      mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
      mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    } else {
      for (int j = 0; j < 4; ++j)
        if (e + j < sg.len) { float v = base[e + j]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    }
  }
  block_minmax(mn, mx, lds);
  if (threadIdx.x == 0 && mn <= mx) {
    atomicMin(&sl[0], enc_f32(mn));
    atomicMin(&sl[1], ~enc_f32(mx));
  }
}

extern "C" int pf_seg_minmax(const float* w_flat, const PfSeg* segs, const PfBlock* blocks,
                             int n_blocks, uint32_t* slots, void* stream) {
  if (n_blocks <= 0) return 0;
  k_seg_minmax<<<n_blocks, PF_THREADS, 0, (hipStream_t)stream>>>(w_flat, segs, blocks, slots);
  PF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// apply kernels.  One 4096-element chunk per block.  (alpha, beta) of the channels a chunk
// touches are decoded once into LDS; codebooks (NUQ) likewise.
// ---------------------------------------------------------------------------------------------
#define PF_LDS_ROWS 1024

template <typename TO> __device__ __forceinline__ void store4(TO* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float* v) {
  uint2 o;
  o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  *reinterpret_cast<uint2*>(p) = o;
}

// nearest codebook entry, ties -> lowest index (tf.argmin); c[j] at stride cs
__device__ __forceinline__ int nuq_argmin(float xn, const float* __restrict__ c, int k, int cs) {
  int best = 0;
  float bd = fabsf(xn - c[0]);
  for (int j = 1; j < k; ++j) {
    const float d = fabsf(xn - c[(int64_t)j * cs]);
    if (d < bd) { bd = d; best = j; }
  }
  return best;
}

template <typename TO, bool NUQ>
__global__ __launch_bounds__(PF_THREADS) void k_seg_apply(const float* __restrict__ w, TO* __restrict__ qw,
                                                          uint8_t* __restrict__ idx_out,
                                                          const float* __restrict__ codebooks,
                                                          const PfSeg* __restrict__ segs,
                                                          const PfBlock* __restrict__ blocks,
                                                          const uint32_t* __restrict__ slots) {
  __shared__ float s_alpha[PF_LDS_ROWS];
  __shared__ float s_beta[PF_LDS_ROWS];
  __shared__ float s_cb[NUQ ? 4096 : 1];
  const PfBlock b = blocks[blockIdx.x];
  const PfSeg sg = segs[b.seg];
  const float* __restrict__ base = w + sg.offset;
  TO* __restrict__ out = qw + sg.offset;
  const uint32_t* __restrict__ sl = slots + 2 * sg.slot_offset;
  const uint32_t e0 = (uint32_t)b.chunk * PF_CHUNK;
  const uint32_t L = (uint32_t)sg.RS * (uint32_t)sg.I;
  const uint32_t len = (uint32_t)sg.len;
  const float kf = uq_k_of_bits(sg.bits);
  const int kc = NUQ ? (1 << sg.bits) : 0;
  const float* __restrict__ cb = NUQ ? (codebooks + sg.cb_offset) : nullptr;

  // ---- stage per-bucket vectors in LDS
  int row0 = 0, nrows = 1;
  bool in_lds = true;
  if (sg.mode == PF_BUCKET_CHANNEL) {
    row0 = b.row0;          // precomputed on the host (pf plan): e0 / L .. min(e0+4095, len-1) / L
    nrows = b.nrows;
    in_lds = nrows <= PF_LDS_ROWS;
  } else if (sg.mode == PF_BUCKET_SPLIT) {
    in_lds = false;
  }
  if (in_lds) {
    // (fixed-trip, fully unrolled: a runtime-trip staging loop here crashes hipcc 7.2's ISel)
#pragma unroll
    for (int q = 0; q < PF_LDS_ROWS / PF_THREADS; ++q) {
      const int r = threadIdx.x + q * PF_THREADS;
      if (r < nrows) {
        float a, bt;
        slot_alpha_beta(sl + 2 * (row0 + r), a, bt);
        s_alpha[r] = a;
        s_beta[r] = bt;
      }
    }
  }
  bool cb_lds = false;
  if (NUQ) {
    if (sg.mode == PF_BUCKET_TENSOR) {
      cb_lds = true;
      for (int j = threadIdx.x; j < kc; j += PF_THREADS) s_cb[j] = cb[j];
    } else if (sg.mode == PF_BUCKET_CHANNEL && in_lds && nrows * kc <= 4096) {
      cb_lds = true;   // s_cb[r*kc + j] = cb[j*n_bucket + row0 + r]
      for (int t = threadIdx.x; t < nrows * kc; t += PF_THREADS) {
        const int r = t / kc, j = t - r * kc;
        s_cb[t] = cb[(int64_t)j * sg.n_bucket + row0 + r];
      }
    }
  }
  __syncthreads();

#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const uint32_t e = e0 + it * 1024 + threadIdx.x * 4;
    if (e >= len) continue;
    float v[4];
    const bool full = (e + 3 < len);
    if (full) {
      float4 t = *reinterpret_cast<const float4*>(base + e);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      for (int j = 0; j < 4; ++j) v[j] = (e + j < len) ? base[e + j] : 0.0f;
    }
    uint32_t packed_idx = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t ej = e + j;
      float alpha, beta;
      uint32_t bucket = 0;
      if (sg.mode == PF_BUCKET_TENSOR) {
        alpha = s_alpha[0]; beta = s_beta[0];
      } else if (sg.mode == PF_BUCKET_CHANNEL) {
        bucket = ej / L;
        if (bucket > (uint32_t)sg.O - 1) bucket = (uint32_t)sg.O - 1;
        if (in_lds) { alpha = s_alpha[bucket - row0]; beta = s_beta[bucket - row0]; }
        else slot_alpha_beta(sl + 2 * bucket, alpha, beta);
      } else {
        bucket = (ej < len ? hwio_flat_index(sg, ej) : 0u) % (uint32_t)sg.n_bucket;
        slot_alpha_beta(sl + 2 * bucket, alpha, beta);
      }
      if (sg.bits <= 0) {
        // bits == 0: tensor is not quantised (first / last layer): plain cast to the compute dtype
      } else if (!NUQ) {
        v[j] = uq_point(v[j], alpha, beta, kf);
      } else {
        const float xn = (v[j] - beta) / alpha;
        int best;
        float cval;
        if (cb_lds) {
          const float* c = (sg.mode == PF_BUCKET_TENSOR) ? s_cb : (s_cb + (bucket - row0) * kc);
          best = nuq_argmin(xn, c, kc, 1);
          cval = c[best];
        } else {
          const float* c = cb + bucket;
          best = nuq_argmin(xn, c, kc, sg.n_bucket);
          cval = c[(int64_t)best * sg.n_bucket];
        }
        const float s = xn + 1e-6f;
        const float sgn = (s > 0.0f) ? 1.0f : ((s < 0.0f) ? -1.0f : 0.0f);
        const float q = cval * sgn;
        v[j] = alpha * q + beta;
        packed_idx |= ((uint32_t)best) << (8 * j);
      }
    }
    if (full) {
      store4<TO>(out + e, v);
      if (NUQ) *reinterpret_cast<uint32_t*>(idx_out + sg.offset + e) = packed_idx;
    } else {
      for (int j = 0; j < 4; ++j)
        if (e + j < len) {
          store_one<TO>(out + e + j, v[j]);
          if (NUQ) idx_out[sg.offset + e + j] = (uint8_t)((packed_idx >> (8 * j)) & 0xFF);
        }
    }
  }
}

extern "C" int pf_seg_uq_apply(const float* w_flat, void* qw_flat, int out_dtype, const PfSeg* segs,
                               const PfBlock* blocks, int n_blocks, const uint32_t* slots,
                               void* stream) {
  if (n_blocks <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == PF_F32)
    k_seg_apply<float, false><<<n_blocks, PF_THREADS, 0, st>>>(w_flat, (float*)qw_flat, nullptr, nullptr, segs, blocks, slots);
  else if (out_dtype == PF_BF16)
    k_seg_apply<bf16_t, false><<<n_blocks, PF_THREADS, 0, st>>>(w_flat, (bf16_t*)qw_flat, nullptr, nullptr, segs, blocks, slots);
  else return (int)hipErrorInvalidValue;
  PF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pf_seg_nuq_apply(const float* w_flat, void* qw_flat, int out_dtype, uint8_t* idx_flat,
                                const float* codebooks, const PfSeg* segs, const PfBlock* blocks,
                                int n_blocks, const uint32_t* slots, void* stream) {
  if (n_blocks <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == PF_F32)
    k_seg_apply<float, true><<<n_blocks, PF_THREADS, 0, st>>>(w_flat, (float*)qw_flat, idx_flat, codebooks, segs, blocks, slots);
  else if (out_dtype == PF_BF16)
    k_seg_apply<bf16_t, true><<<n_blocks, PF_THREADS, 0, st>>>(w_flat, (bf16_t*)qw_flat, idx_flat, codebooks, segs, blocks, slots);
  else return (int)hipErrorInvalidValue;
  PF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// NUQ codebook gradient: dc[j][b] += alpha_b * g_i  for idx_i == j in bucket b.
// Bit-deterministic across runs and ranks: every term is rounded ONCE to a 2^-36 fixed-point grid and accumulated with
// 64-bit INTEGER atomics (integer addition is associative, so the order in which wavefronts arrive cannot change the
// result -- float atomics made `cluster` / `both` mode runs irreproducible); a second launch converts the sums to
// float32 and adds them to dcodebooks.  Range +-1.3e8 with 1.5e-11 resolution: |alpha * g| terms of a gradient
// scatter-sum sit many orders of magnitude inside both ends.  Per-block LDS bins (per-tensor mode) cut the global
// atomics to k per block.
// ---------------------------------------------------------------------------------------------
#define PF_CB_FX_SCALE 68719476736.0                       // 2^36
__device__ __forceinline__ long long cb_to_fx(float v) { return __double2ll_rn((double)v * PF_CB_FX_SCALE); }

template <typename TG>
__global__ __launch_bounds__(PF_THREADS) void k_seg_nuq_cbgrad(const TG* __restrict__ g,
                                                               const uint8_t* __restrict__ idx,
                                                               unsigned long long* __restrict__ acc,
                                                               const PfSeg* __restrict__ segs,
                                                               const PfBlock* __restrict__ blocks,
                                                               const uint32_t* __restrict__ slots) {
  __shared__ unsigned long long bins[256];
  const PfBlock b = blocks[blockIdx.x];
  const PfSeg sg = segs[b.seg];
  if (sg.bits <= 0) return;                   // tensor is not quantised: it has no codebook
  const TG* __restrict__ gb = g + sg.offset;
  const uint8_t* __restrict__ ib = idx + sg.offset;
  const uint32_t* __restrict__ sl = slots + 2 * sg.slot_offset;
  unsigned long long* __restrict__ dc = acc + sg.cb_offset;
  const int kc = 1 << sg.bits;
  const uint32_t e0 = (uint32_t)b.chunk * PF_CHUNK;
  const uint32_t L = (uint32_t)sg.RS * (uint32_t)sg.I;
  const bool tensor_mode = (sg.mode == PF_BUCKET_TENSOR);
  if (tensor_mode) {
    for (int j = threadIdx.x; j < kc; j += PF_THREADS) bins[j] = 0ull;
    __syncthreads();
  }
  float alpha0 = 0.f, beta0 = 0.f;
  if (tensor_mode) slot_alpha_beta(sl, alpha0, beta0);
  for (int t = threadIdx.x; t < PF_CHUNK; t += PF_THREADS) {
    const uint32_t e = e0 + t;
    if (e >= (uint32_t)sg.len) break;
    const float gv = load_one<TG>(gb + e);
    const int j = ib[e];
    if (tensor_mode) {
      atomicAdd(&bins[j], (unsigned long long)cb_to_fx(alpha0 * gv));
    } else {
      uint32_t bucket;
      if (sg.mode == PF_BUCKET_CHANNEL) bucket = e / L;
      else bucket = hwio_flat_index(sg, e) % (uint32_t)sg.n_bucket;
      float a, bt;
      slot_alpha_beta(sl + 2 * bucket, a, bt);
      atomicAdd(&dc[(int64_t)j * sg.n_bucket + bucket], (unsigned long long)cb_to_fx(a * gv));
    }
  }
  if (tensor_mode) {
    __syncthreads();
    for (int j = threadIdx.x; j < kc; j += PF_THREADS)
      if (bins[j] != 0ull) atomicAdd(&dc[j], bins[j]);
  }
}

__global__ __launch_bounds__(PF_THREADS) void k_cb_fx_to_float(const long long* __restrict__ acc, float* __restrict__ dcb,
                                                               int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * PF_THREADS)
    dcb[i] += (float)((double)acc[i] * (1.0 / PF_CB_FX_SCALE));
}

extern "C" int pf_seg_nuq_codebook_grad(const void* g_flat, int g_dtype, const uint8_t* idx_flat,
                                        float* dcodebooks, int64_t* acc_ws, int64_t n_codebook, const PfSeg* segs,
                                        const PfBlock* blocks, int n_blocks, const uint32_t* slots, void* stream) {
  if (n_blocks <= 0) return 0;
  if (acc_ws == nullptr || n_codebook <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipError_t me = hipMemsetAsync(acc_ws, 0, (size_t)n_codebook * sizeof(int64_t), st);
  if (me != hipSuccess) return (int)me;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(acc_ws);
  if (g_dtype == PF_F32)
    k_seg_nuq_cbgrad<float><<<n_blocks, PF_THREADS, 0, st>>>((const float*)g_flat, idx_flat, acc, segs, blocks, slots);
  else if (g_dtype == PF_BF16)
    k_seg_nuq_cbgrad<bf16_t><<<n_blocks, PF_THREADS, 0, st>>>((const bf16_t*)g_flat, idx_flat, acc, segs, blocks, slots);
  else return (int)hipErrorInvalidValue;
  PF_LAUNCH_CHECK();
  k_cb_fx_to_float<<<pf_grid_for(n_codebook, PF_THREADS), PF_THREADS, 0, st>>>((const long long*)acc_ws, dcodebooks, n_codebook);
  PF_LAUNCH_CHECK();
  return 0;
}
