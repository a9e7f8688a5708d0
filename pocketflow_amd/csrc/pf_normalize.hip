// x_hat = (w - beta) / alpha of ONE weight tensor in storage order: input of the NUQ quantile
// initialiser (learners/nonuniform_quantization/utils.py:349-366), which sorts it once.
#include "pf_common.h"

// normalised weights of one tensor, storage order (input of the quantile initialiser)
__global__ __launch_bounds__(PF_THREADS) void k_seg_normalize(const float* __restrict__ w,
                                                              float* __restrict__ xn,
                                                              const PfSeg* __restrict__ segs,
                                                              int seg_index,
                                                              const uint32_t* __restrict__ slots) {
  const PfSeg* __restrict__ sp = segs + seg_index;
  const int64_t len = sp->len, offset = sp->offset, slot_offset = sp->slot_offset;
  const int mode = sp->mode, RS = sp->RS, I = sp->I, O = sp->O, layout = sp->layout, nb = sp->n_bucket;
  const float* __restrict__ base = w + offset;
  const uint32_t* __restrict__ sl = slots + 2 * slot_offset;
  const uint32_t L = (uint32_t)RS * (uint32_t)I;
  for (uint32_t e = blockIdx.x * PF_THREADS + threadIdx.x; e < (uint32_t)len;
       e += gridDim.x * PF_THREADS) {
    uint32_t bucket = 0;
    if (mode == PF_BUCKET_CHANNEL) {
      bucket = e / L;
    } else if (mode == PF_BUCKET_SPLIT) {
      uint32_t f;
      if (layout == 0) { const uint32_t o = e / L; f = (e - o * L) * (uint32_t)O + o; }
      else { const uint32_t i = e / (uint32_t)RS; f = (e - i * (uint32_t)RS) * (uint32_t)I + i; }
      bucket = f % (uint32_t)nb;
    }
    float a, bt;
    slot_alpha_beta(sl + 2 * bucket, a, bt);
    xn[e] = (base[e] - bt) / a;
  }
}

extern "C" int pf_seg_normalize(const float* w_flat, float* xn_out, const PfSeg* segs, int seg_index,
                                const uint32_t* slots, void* stream) {
  k_seg_normalize<<<PF_MAX_GRID / 4, PF_THREADS, 0, (hipStream_t)stream>>>(w_flat, xn_out, segs, seg_index, slots);
  PF_LAUNCH_CHECK();
  return 0;
}
