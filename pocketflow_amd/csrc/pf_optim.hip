// K8 + K14: fused (L2-coupled, masked) Adam / Momentum updates over one flat parameter buffer.
// One launch updates every trainable tensor of the model; HBM-bound (Adam: 28 B/param fp32).
//
// Reference semantics restated here (paths under /root/reference; optimiser maths is TF's
// ApplyAdam / ApplyMomentum kernels, third party):
//   learners/uniform_quantization/learner.py:244-253      Adam(lr) on d(loss)/d(var)
//   learners/weight_sparsification/learner.py:201-212,314-332   Momentum on grad * mask
//   learners/channel_pruning/learner.py:357-368,406-419   Adam | Momentum on grad * mask
//   nets/resnet_at_ilsvrc12.py:132-135    loss += loss_w_dcy * sum l2_loss(v)  => grad += wd * v
#include "pf_common.h"

template <typename TG> __device__ __forceinline__ void load4g(const TG* p, float* g);
template <> __device__ __forceinline__ void load4g<float>(const float* p, float* g) {
  float4 v = *reinterpret_cast<const float4*>(p);
  g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w;
}
template <> __device__ __forceinline__ void load4g<bf16_t>(const bf16_t* p, float* g) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  g[0] = __uint_as_float(v.x << 16); g[1] = __uint_as_float(v.x & 0xFFFF0000u);
  g[2] = __uint_as_float(v.y << 16); g[3] = __uint_as_float(v.y & 0xFFFF0000u);
}

__device__ __forceinline__ float eff_grad(float g, float p, float mask, bool decay, float wd,
                                          float g_scale) {
  float ge = g * g_scale;
  if (decay) ge = ge + wd * p;
  return ge * mask;
}

template <typename TG, bool MASK>
__global__ __launch_bounds__(PF_THREADS) void k_adam_flat(float* __restrict__ p, const TG* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          const float* __restrict__ mask, int64_t n,
                                                          int64_t n_decay, float wd, float g_scale,
                                                          float alpha_t, float omb1, float omb2,
                                                          float eps, const float* __restrict__ hp) {
  // hp (captured steps): the step's alpha_t is read from device memory -- a launch recorded in a hipGraph keeps its BY-VALUE
  // arguments for ever, and alpha_t (learning-rate schedule, bias correction) changes every step
  if (hp != nullptr) alpha_t = hp[0];
  const int64_t nv = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; i < nv;
       i += (int64_t)gridDim.x * PF_THREADS) {
    const int64_t e = i << 2;
    float gg[4], mk[4] = {1.f, 1.f, 1.f, 1.f};
    load4g<TG>(g + e, gg);
    float4 pp = *reinterpret_cast<float4*>(p + e);
    float4 mm = *reinterpret_cast<float4*>(m + e);
    float4 vv = *reinterpret_cast<float4*>(v + e);
    if (MASK) { float4 t = *reinterpret_cast<const float4*>(mask + e); mk[0] = t.x; mk[1] = t.y; mk[2] = t.z; mk[3] = t.w; }
    float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ge = eff_grad(gg[j], pa[j], mk[j], (e + j) < n_decay, wd, g_scale);
      ma[j] = ma[j] + (ge - ma[j]) * omb1;
      va[j] = va[j] + (ge * ge - va[j]) * omb2;
      pa[j] = pa[j] - (ma[j] * alpha_t) / (sqrtf(va[j]) + eps);
    }
    *reinterpret_cast<float4*>(p + e) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    *reinterpret_cast<float4*>(m + e) = make_float4(ma[0], ma[1], ma[2], ma[3]);
    *reinterpret_cast<float4*>(v + e) = make_float4(va[0], va[1], va[2], va[3]);
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0) {
    for (int64_t e = (nv << 2) + threadIdx.x; e < n; e += PF_THREADS) {
      const float mk = MASK ? mask[e] : 1.0f;
      const float ge = eff_grad(load_one<TG>(g + e), p[e], mk, e < n_decay, wd, g_scale);
      const float mn = m[e] + (ge - m[e]) * omb1;
      const float vn = v[e] + (ge * ge - v[e]) * omb2;
      m[e] = mn; v[e] = vn;
      p[e] = p[e] - (mn * alpha_t) / (sqrtf(vn) + eps);
    }
  }
}

static int adam_launch(float* p, const void* g, int g_dtype, float* m, float* v,
                       const float* mask, int64_t n, int64_t n_decay, float wd, float g_scale,
                       float lr, float beta1, float beta2, float eps, float beta1_power,
                       float beta2_power, const float* hp, void* stream) {
  if (n <= 0) return 0;
  if (!pf_aligned16(p) || !pf_aligned16(g) || !pf_aligned16(m) || !pf_aligned16(v) ||
      (mask && !pf_aligned16(mask)))
    return (int)hipErrorInvalidValue;
  // TF ApplyAdam: alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power), all float32
  const float alpha_t = lr * sqrtf(1.0f - beta2_power) / (1.0f - beta1_power);
  const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
  hipStream_t st = (hipStream_t)stream;
  const int grid = pf_grid_for(n, PF_THREADS * 4);
#define PF_AD(TG, MK) k_adam_flat<TG, MK><<<grid, PF_THREADS, 0, st>>>(p, (const TG*)g, m, v, mask, n, n_decay, wd, g_scale, alpha_t, omb1, omb2, eps, hp)
  if (g_dtype == PF_F32) { if (mask) PF_AD(float, true); else PF_AD(float, false); }
  else if (g_dtype == PF_BF16) { if (mask) PF_AD(bf16_t, true); else PF_AD(bf16_t, false); }
  else return (int)hipErrorInvalidValue;
#undef PF_AD
  PF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pf_adam_flat(float* p, const void* g, int g_dtype, float* m, float* v,
                            const float* mask, int64_t n, int64_t n_decay, float wd, float g_scale,
                            float lr, float beta1, float beta2, float eps, float beta1_power,
                            float beta2_power, void* stream) {
  return adam_launch(p, g, g_dtype, m, v, mask, n, n_decay, wd, g_scale, lr, beta1, beta2, eps, beta1_power, beta2_power, nullptr,
                     stream);
}

// The same update with alpha_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) read from hp[0] (device memory, written by pf_set_floats
// before the step): the form a step captured in a hipGraph launches.
extern "C" int pf_adam_flat_dev(float* p, const void* g, int g_dtype, float* m, float* v,
                                const float* mask, int64_t n, int64_t n_decay, float wd, float g_scale,
                                const float* hp, float beta1, float beta2, float eps, void* stream) {
  if (hp == nullptr) return (int)hipErrorInvalidValue;
  return adam_launch(p, g, g_dtype, m, v, mask, n, n_decay, wd, g_scale, 0.f, beta1, beta2, eps, 0.5f, 0.5f, hp, stream);
}

template <typename TG, bool MASK>
__global__ __launch_bounds__(PF_THREADS) void k_momentum_flat(float* __restrict__ p, const TG* __restrict__ g,
                                                              float* __restrict__ acc,
                                                              const float* __restrict__ mask, int64_t n,
                                                              int64_t n_decay, float wd, float g_scale,
                                                              float lr, float mu, const float* __restrict__ hp) {
  if (hp != nullptr) lr = hp[1];                        // captured steps: the learning rate lives in device memory
  const int64_t nv = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; i < nv;
       i += (int64_t)gridDim.x * PF_THREADS) {
    const int64_t e = i << 2;
    float gg[4], mk[4] = {1.f, 1.f, 1.f, 1.f};
    load4g<TG>(g + e, gg);
    float4 pp = *reinterpret_cast<float4*>(p + e);
    float4 aa = *reinterpret_cast<float4*>(acc + e);
    if (MASK) { float4 t = *reinterpret_cast<const float4*>(mask + e); mk[0] = t.x; mk[1] = t.y; mk[2] = t.z; mk[3] = t.w; }
    float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ac[4] = {aa.x, aa.y, aa.z, aa.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ge = eff_grad(gg[j], pa[j], mk[j], (e + j) < n_decay, wd, g_scale);
      ac[j] = mu * ac[j] + ge;            // acc <- mu*acc + g
      pa[j] = pa[j] - lr * ac[j];         // var <- var - lr*acc   (no Nesterov)
    }
    *reinterpret_cast<float4*>(p + e) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    *reinterpret_cast<float4*>(acc + e) = make_float4(ac[0], ac[1], ac[2], ac[3]);
  }
  if (blockIdx.x == 0) {
    for (int64_t e = (nv << 2) + threadIdx.x; e < n; e += PF_THREADS) {
      const float mk = MASK ? mask[e] : 1.0f;
      const float ge = eff_grad(load_one<TG>(g + e), p[e], mk, e < n_decay, wd, g_scale);
      const float a = mu * acc[e] + ge;
      acc[e] = a;
      p[e] = p[e] - lr * a;
    }
  }
}

static int momentum_launch(float* p, const void* g, int g_dtype, float* acc, const float* mask,
                           int64_t n, int64_t n_decay, float wd, float g_scale, float lr,
                           float momentum, const float* hp, void* stream) {
  if (n <= 0) return 0;
  if (!pf_aligned16(p) || !pf_aligned16(g) || !pf_aligned16(acc) || (mask && !pf_aligned16(mask)))
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int grid = pf_grid_for(n, PF_THREADS * 4);
#define PF_MO(TG, MK) k_momentum_flat<TG, MK><<<grid, PF_THREADS, 0, st>>>(p, (const TG*)g, acc, mask, n, n_decay, wd, g_scale, lr, momentum, hp)
  if (g_dtype == PF_F32) { if (mask) PF_MO(float, true); else PF_MO(float, false); }
  else if (g_dtype == PF_BF16) { if (mask) PF_MO(bf16_t, true); else PF_MO(bf16_t, false); }
  else return (int)hipErrorInvalidValue;
#undef PF_MO
  PF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pf_momentum_flat(float* p, const void* g, int g_dtype, float* acc, const float* mask,
                                int64_t n, int64_t n_decay, float wd, float g_scale, float lr,
                                float momentum, void* stream) {
  return momentum_launch(p, g, g_dtype, acc, mask, n, n_decay, wd, g_scale, lr, momentum, nullptr, stream);
}

// ... with the learning rate read from hp[1] (captured steps, see pf_adam_flat_dev)
extern "C" int pf_momentum_flat_dev(float* p, const void* g, int g_dtype, float* acc, const float* mask,
                                    int64_t n, int64_t n_decay, float wd, float g_scale, const float* hp,
                                    float momentum, void* stream) {
  if (hp == nullptr) return (int)hipErrorInvalidValue;
  return momentum_launch(p, g, g_dtype, acc, mask, n, n_decay, wd, g_scale, 0.f, momentum, hp, stream);
}

// dst[0..3] = (a, b, c, d): per-step scalars of a captured step.  The values travel as kernel arguments (copied at launch time), so
// the host may run any number of steps ahead of the GPU -- a pinned staging buffer would be overwritten before its copy executes.
__global__ void k_set_floats(float* __restrict__ dst, float a, float b, float c, float d) {
  if (threadIdx.x == 0) { dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d; }
}

extern "C" int pf_set_floats(float* dst, float a, float b, float c, float d, void* stream) {
  if (dst == nullptr) return (int)hipErrorInvalidValue;
  k_set_floats<<<1, 64, 0, (hipStream_t)stream>>>(dst, a, b, c, d);
  PF_LAUNCH_CHECK();
  return 0;
}
