// R x S convolutions with FEW input channels (16, 32, 48: ResNet-20 @ CIFAR-10, utils/external/resnet_model.py:156-199 building blocks with
// 16 / 32 / 64 filters, nets/resnet_at_cifar10.py:49-62) on the in-tree 1x1 kernels.  The implicit-GEMM kernel of pf_igemm.hip takes its
// k-steps as (tap, 64 channels); below 64 channels a k-step would span several taps of several pixels.  These layers are small (4 MB
// of activations at batch 128), so the gather is done ONCE, in HBM, by a streaming kernel:
//     Xcol[m][(r*S + s)*C + c] = X[img][ho*st + r - pad_h][wo*st + s - pad_w][c]          (zero outside the image)
// and the convolution, its backward-data and its backward-filter become the 1x1 products Y = Xcol W^T, dXcol = dY W, dW = dY^T Xcol of
// pf_conv.hip / pf_wrw.hip (W[N][R][S][C] IS the [N][R*S*C] matrix, no copy), followed -- backward-data -- by the inverse gather
//     dX[p][c] = sum over the taps (r, s) whose output position (ho, wo) = ((h + pad_h - r) / st, (w + pad_w - s) / st) exists
//                of dXcol[(img, ho, wo)][(r*S + s)*C + c]
// (a gather per input pixel, float32 sum of <= R*S terms in tap order, one rounding: deterministic, no atomics).  Until round 4
// these layers ran in MIOpen.  bf16, C % 8 == 0 (16-byte vectors).
#include "pf_common.h"

#define I2C_T 256

__global__ __launch_bounds__(I2C_T) void k_im2col(const bf16_t* __restrict__ x, bf16_t* __restrict__ xcol, int imgs, int H, int W, int C,
                                                  int R, int S, int stride, int pad_h, int pad_w, int Ho, int Wo, int64_t total) {
  const int vec = C >> 3, RS = R * S;
  for (int64_t i = (int64_t)blockIdx.x * I2C_T + threadIdx.x; i < total; i += (int64_t)gridDim.x * I2C_T) {
    const int v = (int)(i % vec);
    const int64_t q = i / vec;
    const int t = (int)(q % RS);
    const int64_t m = q / RS;
    const int wo = (int)(m % Wo);
    const int64_t q2 = m / Wo;
    const int ho = (int)(q2 % Ho), img = (int)(q2 / Ho);
    const int r = t / S, s = t - r * S;
    const int h = ho * stride + r - pad_h, w = wo * stride + s - pad_w;
    uint4 val = make_uint4(0, 0, 0, 0);
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W)
      val = *reinterpret_cast<const uint4*>(x + ((int64_t)(img * H + h) * W + w) * C + v * 8);
    *reinterpret_cast<uint4*>(xcol + (m * RS + t) * C + v * 8) = val;
  }
}

__global__ __launch_bounds__(I2C_T) void k_col2im(const bf16_t* __restrict__ dxcol, bf16_t* __restrict__ dx, int imgs, int H, int W, int C,
                                                  int R, int S, int stride, int pad_h, int pad_w, int Ho, int Wo, int64_t total) {
  const int vec = C >> 3, RS = R * S;
  for (int64_t i = (int64_t)blockIdx.x * I2C_T + threadIdx.x; i < total; i += (int64_t)gridDim.x * I2C_T) {
    const int v = (int)(i % vec);
    const int64_t p = i / vec;
    const int w = (int)(p % W);
    const int64_t q2 = p / W;
    const int h = (int)(q2 % H), img = (int)(q2 / H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int r = 0; r < R; ++r) {
      const int th = h + pad_h - r;
      if (th < 0) continue;
      const int ho = th / stride;
      if (ho * stride != th || ho >= Ho) continue;
      for (int s = 0; s < S; ++s) {
        const int tw = w + pad_w - s;
        if (tw < 0) continue;
        const int wo = tw / stride;
        if (wo * stride != tw || wo >= Wo) continue;
        float f[8];
        load_vec<bf16_t>(dxcol + (((int64_t)(img * Ho + ho) * Wo + wo) * RS + r * S + s) * C + v * 8, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = acc[j] + f[j];
      }
    }
    store_vec<bf16_t>(dx + p * C + v * 8, acc);
  }
}

static bool i2c_ok(const void* a, const void* b, int imgs, int H, int W, int C, int R, int S, int stride, int pad_h, int pad_w, int Ho, int Wo) {
  return a != nullptr && b != nullptr && pf_aligned16(a) && pf_aligned16(b) && imgs > 0 && H > 0 && W > 0 && C > 0 && (C % 8) == 0 && R > 0 &&
         S > 0 && stride > 0 && pad_h >= 0 && pad_w >= 0 && Ho > 0 && Wo > 0 && (int64_t)imgs * Ho * Wo * R * S * C < ((int64_t)1 << 40);
}

// xcol [imgs*Ho*Wo][R*S*C] from x [imgs][H][W][C]
extern "C" int pf_im2col(const void* x, void* xcol, int imgs, int H, int W, int C, int R, int S, int stride, int pad_h, int pad_w, int Ho,
                         int Wo, void* stream) {
  if (!i2c_ok(x, xcol, imgs, H, W, C, R, S, stride, pad_h, pad_w, Ho, Wo)) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)imgs * Ho * Wo * R * S * (C / 8);
  k_im2col<<<pf_grid_for(total, I2C_T), I2C_T, 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)xcol, imgs, H, W, C, R, S, stride, pad_h,
                                                                        pad_w, Ho, Wo, total);
  PF_LAUNCH_CHECK();
  return 0;
}

// dx [imgs][H][W][C] from dxcol [imgs*Ho*Wo][R*S*C]
extern "C" int pf_col2im(const void* dxcol, void* dx, int imgs, int H, int W, int C, int R, int S, int stride, int pad_h, int pad_w, int Ho,
                         int Wo, void* stream) {
  if (!i2c_ok(dxcol, dx, imgs, H, W, C, R, S, stride, pad_h, pad_w, Ho, Wo)) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)imgs * H * W * (C / 8);
  k_col2im<<<pf_grid_for(total, I2C_T), I2C_T, 0, (hipStream_t)stream>>>((const bf16_t*)dxcol, (bf16_t*)dx, imgs, H, W, C, R, S, stride, pad_h,
                                                                        pad_w, Ho, Wo, total);
  PF_LAUNCH_CHECK();
  return 0;
}
