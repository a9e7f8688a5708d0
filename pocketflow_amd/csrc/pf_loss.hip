// K10 + K11: hard-label softmax cross-entropy and the distillation soft-label cross-entropy,
// forward AND backward in one row-wise kernel (one 256-thread workgroup per example; row
// reductions by 64-lane shuffles + 4-entry LDS).  Fixed summation order => bit-deterministic.
//
// Reference semantics restated here (paths under /root/reference):
//   nets/resnet_at_ilsvrc12.py:132   loss = tf.losses.softmax_cross_entropy(labels, outputs)
//   learners/distillation_helper.py:98-100
//       logits_soft = logits_pri / T ; labels_soft = softmax(logits_dst / T)
//       loss = loss_w_dst * tf.losses.softmax_cross_entropy(labels_soft, logits_soft)
//   (tf.losses.softmax_cross_entropy: per-example -sum_c l_c log_softmax(z)_c, mean over batch)
#include "pf_common.h"

__device__ __forceinline__ float block_max(float v, float* lds) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) lds[w] = v;
  __syncthreads();
  v = fmaxf(fmaxf(lds[0], lds[1]), fmaxf(lds[2], lds[3]));
  __syncthreads();
  return v;
}

template <typename TS, typename TT, typename TD, bool DST>
__global__ __launch_bounds__(PF_THREADS) void k_ce_distill(const TS* __restrict__ z_s,
                                                           const float* __restrict__ labels,
                                                           const TT* __restrict__ z_t, int B, int C,
                                                           float T, float loss_w,
                                                           TD* __restrict__ dz, float* __restrict__ row_ws) {
  __shared__ float lds[4];
  const int b = blockIdx.x;
  const TS* __restrict__ zs = z_s + (int64_t)b * C;
  const float* __restrict__ lab = labels + (int64_t)b * C;
  const TT* __restrict__ zt = DST ? (z_t + (int64_t)b * C) : nullptr;
  TD* __restrict__ d = dz + (int64_t)b * C;

  // pass 1: row maxima
  float ms = -INFINITY, mt = -INFINITY;
  for (int c = threadIdx.x; c < C; c += PF_THREADS) {
    ms = fmaxf(ms, load_one<TS>(zs + c));
    if (DST) mt = fmaxf(mt, load_one<TT>(zt + c));
  }
  ms = block_max(ms, lds);
  if (DST) mt = block_max(mt, lds);
  const float msT = ms / T, mtT = mt / T;

  // pass 2: sums of exponentials (hard: z_s ; soft: z_s/T and z_t/T) and sum of labels
  float se = 0.f, seT = 0.f, stT = 0.f, sl = 0.f;
  for (int c = threadIdx.x; c < C; c += PF_THREADS) {
    const float z = load_one<TS>(zs + c);
    se += expf(z - ms);
    sl += lab[c];
    if (DST) {
      seT += expf(z / T - msT);
      stT += expf(load_one<TT>(zt + c) / T - mtT);
    }
  }
  se = block_sum(se, lds);
  sl = block_sum(sl, lds);
  if (DST) { seT = block_sum(seT, lds); stT = block_sum(stT, lds); }
  const float lse = logf(se);
  const float lseT = DST ? logf(seT) : 0.f;

  // pass 3: per-example losses + dlogits
  const float invB = 1.0f / (float)B;
  const float wBT = DST ? (loss_w / ((float)B * T)) : 0.f;
  float ce = 0.f, ced = 0.f;
  for (int c = threadIdx.x; c < C; c += PF_THREADS) {
    const float z = load_one<TS>(zs + c);
    const float l = lab[c];
    const float lsm = (z - ms) - lse;                 // log_softmax(z_s)_c
    ce -= l * lsm;
    float g = (expf(lsm) * sl - l) * invB;
    if (DST) {
      const float lsmT = (z / T - msT) - lseT;        // log_softmax(z_s / T)_c
      const float pt = expf(load_one<TT>(zt + c) / T - mtT) / stT;   // softmax(z_t / T)_c
      ced -= pt * lsmT;
      g += wBT * (expf(lsmT) - pt);
    }
    store_one<TD>(d + c, g);
  }
  ce = block_sum(ce, lds);
  if (DST) ced = block_sum(ced, lds);
  if (threadIdx.x == 0) { row_ws[2 * b] = ce; row_ws[2 * b + 1] = ced; }
}

__global__ __launch_bounds__(PF_THREADS) void k_loss_finalize(const float* __restrict__ row_ws, int B,
                                                              float loss_w, float* __restrict__ losses) {
  __shared__ float lds[4];
  float a = 0.f, d = 0.f;
  for (int b = threadIdx.x; b < B; b += PF_THREADS) { a += row_ws[2 * b]; d += row_ws[2 * b + 1]; }
  a = block_sum(a, lds);
  d = block_sum(d, lds);
  if (threadIdx.x == 0) {
    losses[0] = a / (float)B;
    losses[1] = loss_w * (d / (float)B);
  }
}

extern "C" int pf_ce_distill_fwd_bwd(const void* z_s, int zs_dtype, const float* labels,
                                     const void* z_t, int zt_dtype, int B, int C, float tempr,
                                     float loss_w, float* losses, void* dz_s, int dz_dtype,
                                     float* row_ws, void* stream) {
  if (B <= 0 || C <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
#define PF_CE(TS, TT, TD)                                                                                   \
  do {                                                                                                      \
    if (z_t) k_ce_distill<TS, TT, TD, true><<<B, PF_THREADS, 0, st>>>((const TS*)z_s, labels, (const TT*)z_t, B, C, tempr, loss_w, (TD*)dz_s, row_ws); \
    else k_ce_distill<TS, TT, TD, false><<<B, PF_THREADS, 0, st>>>((const TS*)z_s, labels, (const TT*)nullptr, B, C, tempr, loss_w, (TD*)dz_s, row_ws); \
  } while (0)
  const int key = zs_dtype * 4 + (z_t ? zt_dtype : zs_dtype) * 2 + dz_dtype;
  switch (key) {
    case 0: PF_CE(float, float, float); break;
    case 1: PF_CE(float, float, bf16_t); break;
    case 2: PF_CE(float, bf16_t, float); break;
    case 3: PF_CE(float, bf16_t, bf16_t); break;
    case 4: PF_CE(bf16_t, float, float); break;
    case 5: PF_CE(bf16_t, float, bf16_t); break;
    case 6: PF_CE(bf16_t, bf16_t, float); break;
    case 7: PF_CE(bf16_t, bf16_t, bf16_t); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef PF_CE
  k_loss_finalize<<<1, PF_THREADS, 0, st>>>(row_ws, B, z_t ? loss_w : 0.0f, losses);
  PF_LAUNCH_CHECK();
  return 0;
}
