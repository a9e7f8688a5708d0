// K13: input pipeline tail on the device -- bilinear resize (TensorFlow-1.x semantics) + horizontal flip +
// central-crop window + per-channel mean subtraction + cast, one launch per mini-batch.
//
// Replaces, per image (utils/external/imagenet_preprocessing.py:226-260 of the reference):
//   training:  random_flip_left_right(crop) -> resize_images(BILINEAR, align_corners=False) -> image - means
//   eval:      resize_images(shorter side 256) -> central crop 224x224 -> image - means
// The host decodes the JPEG (and, for training, cuts the sampled crop window); the variable-size uint8 HWC
// images of a batch are packed back to back in one staging buffer described by a PfImageDesc table.
//
// Legacy TF bilinear kernel (resize_bilinear_op.cc, half_pixel_centers = false):
//   in = out_index * scale, scale = in_size / (float)out_size;  lo = floor(in), hi = min(ceil(in), in_size - 1),
//   lerp = in - lo;  top = tl + (tr - tl) * xl;  bottom = bl + (br - bl) * xl;  out = top + (bottom - top) * yl
// in float32 with one rounding per operation (library is built with -ffp-contract=off).
// The flip acts on the SOURCE (the reference flips before it resizes; the legacy kernel is not mirror symmetric).
//
// Memory-bound and tiny next to the network: 224*224*3 outputs per image, 12 source bytes per output pixel.
#include "pf_common.h"

template <typename T>
__global__ __launch_bounds__(PF_THREADS) void k_image_resize(const uint8_t* __restrict__ src,
                                                              const PfImageDesc* __restrict__ desc,
                                                              T* __restrict__ out, int OH, int OW, float m0, float m1,
                                                              float m2) {
  const PfImageDesc d = desc[blockIdx.y];
  const uint8_t* __restrict__ img = src + d.offset;
  const int npix = OH * OW;
  T* __restrict__ o = out + (int64_t)blockIdx.y * npix * 3;
  for (int p = blockIdx.x * PF_THREADS + threadIdx.x; p < npix; p += gridDim.x * PF_THREADS) {
    const int y = p / OW, x = p - y * OW;
    const float in_y = (float)(y + d.off_y) * d.scale_y;
    const float in_x = (float)(x + d.off_x) * d.scale_x;
    const float fy = floorf(in_y), fx = floorf(in_x);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min((int)ceilf(in_y), d.h - 1), x1 = min((int)ceilf(in_x), d.w - 1);
    const float ly = in_y - fy, lx = in_x - fx;
    const int c0 = d.flip ? d.w - 1 - x0 : x0;
    const int c1 = d.flip ? d.w - 1 - x1 : x1;
    const uint8_t* r0 = img + (int64_t)y0 * d.w * 3;
    const uint8_t* r1 = img + (int64_t)y1 * d.w * 3;
    const float mean[3] = {m0, m1, m2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tl = (float)r0[c0 * 3 + c], tr = (float)r0[c1 * 3 + c];
      const float bl = (float)r1[c0 * 3 + c], br = (float)r1[c1 * 3 + c];
      const float top = tl + (tr - tl) * lx;
      const float bot = bl + (br - bl) * lx;
      store_one<T>(o + (int64_t)p * 3 + c, (top + (bot - top) * ly) - mean[c]);
    }
  }
}

extern "C" int pf_image_resize_bilinear(const void* src, const PfImageDesc* desc, void* out, int out_dtype, int B,
                                        int OH, int OW, float mean_r, float mean_g, float mean_b, void* stream) {
  if (B <= 0 || OH <= 0 || OW <= 0 || src == nullptr || desc == nullptr || out == nullptr) return (int)hipErrorInvalidValue;
  const int npix = OH * OW;
  dim3 grid((unsigned)min((npix + PF_THREADS - 1) / PF_THREADS, 64), (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == PF_F32)
    k_image_resize<float><<<grid, PF_THREADS, 0, st>>>((const uint8_t*)src, desc, (float*)out, OH, OW, mean_r, mean_g, mean_b);
  else if (out_dtype == PF_BF16)
    k_image_resize<bf16_t><<<grid, PF_THREADS, 0, st>>>((const uint8_t*)src, desc, (bf16_t*)out, OH, OW, mean_r, mean_g, mean_b);
  else
    return (int)hipErrorInvalidValue;
  PF_LAUNCH_CHECK();
  return 0;
}
