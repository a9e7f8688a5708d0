// K13: the stem's max-pooling (tf.layers.max_pooling2d, padding SAME -- utils/external/resnet_model.py:522-526) and its
// gradient (MaxPoolGrad) on NHWC tensors.  HBM-bound: one lane = 8 consecutive channels (16 bytes) of one pixel.
//   forward : y = max over the window (padding = -inf, i.e. clipped windows); writes the position of the FIRST maximum
//             in window order (r * k + s) as one byte per element -- what the gradient routes to (TF's and torch's rule)
//   backward: gather form -- every INPUT pixel collects dy from the <= ceil(k/stride)^2 windows that contain it and
//             selected it; no atomics, every dx element is written exactly once (also the zeros), deterministic.
// Algorithmic bytes per output element (k = 3, stride 2, bf16): forward 4*2 (input, read once through L2) + 2 + 1,
// backward 4*2 (dx) + 2 + 1.
#include "pf_common.h"
#include <stdlib.h>

template <typename T>
__global__ __launch_bounds__(PF_THREADS) void k_maxpool_fwd(const T* __restrict__ x, T* __restrict__ y,
                                                            uint8_t* __restrict__ idx, int B, int H, int W, int C,
                                                            int k, int stride, int pad_h, int pad_w, int Ho, int Wo) {
  const int CV = C >> 3;
  const int64_t total = (int64_t)B * Ho * Wo * CV;
  for (int64_t e = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; e < total; e += (int64_t)gridDim.x * PF_THREADS) {
    const int cv = (int)(e % CV);
    int64_t p = e / CV;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float best[8];
    int arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; arg[j] = 0; }
    for (int r = 0; r < k; ++r) {
      const int hi = ho * stride + r - pad_h;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int s = 0; s < k; ++s) {
        const int wi = wo * stride + s - pad_w;
        if ((unsigned)wi >= (unsigned)W) continue;
        float v[8];
        load8<T>(x + (((int64_t)b * H + hi) * W + wi) * C + (cv << 3), v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; arg[j] = r * k + s; }
      }
    }
    const int64_t o = (((int64_t)b * Ho + ho) * Wo + wo) * C + (cv << 3);
    store8<T>(y + o, best);
    if (idx != nullptr) {
      uint2 pk;
      pk.x = (uint32_t)arg[0] | ((uint32_t)arg[1] << 8) | ((uint32_t)arg[2] << 16) | ((uint32_t)arg[3] << 24);
      pk.y = (uint32_t)arg[4] | ((uint32_t)arg[5] << 8) | ((uint32_t)arg[6] << 16) | ((uint32_t)arg[7] << 24);
      *reinterpret_cast<uint2*>(idx + o) = pk;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(PF_THREADS) void k_maxpool_bwd(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                            T* __restrict__ dx, int B, int H, int W, int C, int k,
                                                            int stride, int pad_h, int pad_w, int Ho, int Wo) {
  const int CV = C >> 3;
  const int64_t total = (int64_t)B * H * W * CV;
  for (int64_t e = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; e < total; e += (int64_t)gridDim.x * PF_THREADS) {
    const int cv = (int)(e % CV);
    int64_t p = e / CV;
    const int wi = (int)(p % W); p /= W;
    const int hi = (int)(p % H);
    const int b = (int)(p / H);
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    // windows containing (hi, wi): ho*stride - pad_h <= hi <= ho*stride - pad_h + k - 1
    const int ho_hi = (hi + pad_h) / stride;
    const int wo_hi = (wi + pad_w) / stride;
    for (int ho = ho_hi; ho >= 0 && ho * stride - pad_h + k - 1 >= hi; --ho) {
      if (ho >= Ho) continue;
      const int r = hi - (ho * stride - pad_h);
      for (int wo = wo_hi; wo >= 0 && wo * stride - pad_w + k - 1 >= wi; --wo) {
        if (wo >= Wo) continue;
        const int s = wi - (wo * stride - pad_w);
        const int64_t o = (((int64_t)b * Ho + ho) * Wo + wo) * C + (cv << 3);
        const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
        float d[8];
        load8<T>(dy + o, d);
        const uint32_t want = (uint32_t)(r * k + s);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t a = ((j < 4 ? pk.x : pk.y) >> ((j & 3) * 8)) & 0xFFu;
          if (a == want) g[j] += d[j];
        }
      }
    }
    store8<T>(dx + (((int64_t)b * H + hi) * W + wi) * C + (cv << 3), g);
  }
}

// ---- 3x3 / stride 2 (the ResNet stem), round 3 -----------------------------------------------------------------------------
// Same results as the generic kernels above, element for element (same window order, same NaN rule, same summation order
// of the gradient), without their per-element 64-bit divisions and data-dependent loops: every load of a thread is issued
// before the first comparison.  forward: one lane = one output pixel x 8 channels, nine 16-byte loads (clipped taps read
// nothing and count as -inf).  backward: one lane = a 2 x 2 block of INPUT pixels x 8 channels -- the block touches exactly the
// four windows (a-1..a) x (b-1..b), so 4 (dy, index) vectors serve 4 pixels (the per-pixel gather loads 9 for them).
template <typename T>
__global__ __launch_bounds__(PF_THREADS) void k_maxpool3s2_fwd(const T* __restrict__ x, T* __restrict__ y,
                                                               uint8_t* __restrict__ idx, int B, int H, int W, int C,
                                                               int pad_h, int pad_w, int Ho, int Wo) {
  const int CV = C >> 3;
  const int total = B * Ho * Wo * CV;
  for (int64_t e64 = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; e64 < total; e64 += (int64_t)gridDim.x * PF_THREADS) {
    const int e = (int)e64;                                              // total < 2^31 (launcher): 32-bit index arithmetic
    const int cv = e % CV;
    int p = e / CV;
    const int wo = p % Wo; p /= Wo;
    const int ho = p % Ho;
    const int b = p / Ho;
    const int h0 = ho * 2 - pad_h, w0 = wo * 2 - pad_w;
    float v[9][8];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int hi = h0 + r, wi = w0 + s;
        // UNCONDITIONAL load from the clamped position, the clipped tap masked afterwards (round 5): written as `if (inside) load
        // else -inf`, hipcc branches around every load and waits vmcnt(0) behind it -- nine dependent round trips per output pixel
        // (228 us per launch at 256 x 112 x 112 x 64 against a 85 us HBM floor)
        const int hc = hi < 0 ? 0 : (hi >= H ? H - 1 : hi), wc = wi < 0 ? 0 : (wi >= W ? W - 1 : wi);
        load8<T>(x + (((int64_t)b * H + hc) * W + wc) * C + (cv << 3), v[r * 3 + s]);
      }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const bool ok = (unsigned)(h0 + r) < (unsigned)H && (unsigned)(w0 + s) < (unsigned)W;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[r * 3 + s][j] = ok ? v[r * 3 + s][j] : -INFINITY;   // never selected: -inf > best is false
      }
    float best[8];
    int arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; arg[j] = 0; }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (v[t][j] > best[j] || v[t][j] != v[t][j]) { best[j] = v[t][j]; arg[j] = t; }
    const int64_t o = (((int64_t)b * Ho + ho) * Wo + wo) * C + (cv << 3);
    store8<T>(y + o, best);
    if (idx != nullptr) {
      uint2 pk;
      pk.x = (uint32_t)arg[0] | ((uint32_t)arg[1] << 8) | ((uint32_t)arg[2] << 16) | ((uint32_t)arg[3] << 24);
      pk.y = (uint32_t)arg[4] | ((uint32_t)arg[5] << 8) | ((uint32_t)arg[6] << 16) | ((uint32_t)arg[7] << 24);
      *reinterpret_cast<uint2*>(idx + o) = pk;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(PF_THREADS) void k_maxpool3s2_bwd(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                               T* __restrict__ dx, int B, int H, int W, int C, int pad_h,
                                                               int pad_w, int Ho, int Wo, int Ha, int Wb) {
  const int CV = C >> 3;
  const int total = B * Ha * Wb * CV;
  for (int64_t e64 = (int64_t)blockIdx.x * PF_THREADS + threadIdx.x; e64 < total; e64 += (int64_t)gridDim.x * PF_THREADS) {
    const int e = (int)e64;                                              // total < 2^31 (launcher): 32-bit index arithmetic
    const int cv = e % CV;
    int p = e / CV;
    const int bb = p % Wb; p /= Wb;
    const int a = p % Ha;
    const int b = p / Ha;
    // windows (a - i, bb - k), i, k in {0, 1}: padded coordinates u = hi + pad_h in {2a, 2a+1}, window ho covers u in [2ho, 2ho+2]
    float d[2][2][8];
    uint2 pk[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int ho = a - i, wo = bb - k;
        // (unconditional loads from the clamped window, masked afterwards: see the forward kernel)
        const bool ok = (unsigned)ho < (unsigned)Ho && (unsigned)wo < (unsigned)Wo;
        const int hoc = ho < 0 ? 0 : (ho >= Ho ? Ho - 1 : ho), woc = wo < 0 ? 0 : (wo >= Wo ? Wo - 1 : wo);
        const int64_t o = (((int64_t)b * Ho + hoc) * Wo + woc) * C + (cv << 3);
        const uint2 pki = *reinterpret_cast<const uint2*>(idx + o);
        load8<T>(dy + o, d[i][k]);
        pk[i][k] = ok ? pki : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);        // 255: no tap
      }
#pragma unroll
    for (int du = 0; du < 2; ++du) {
      const int hi = 2 * a + du - pad_h;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int dv = 0; dv < 2; ++dv) {
        const int wi = 2 * bb + dv - pad_w;
        if ((unsigned)wi >= (unsigned)W) continue;
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = 0.f;
        // the generic kernel's order: ho descending from the last window that contains the row, wo likewise
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = du + 2 * i;                                        // tap row inside window a - i
          if (r > 2) continue;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int s = dv + 2 * k;
            if (s > 2) continue;
            const uint32_t want = (uint32_t)(r * 3 + s);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const uint32_t t = ((j < 4 ? pk[i][k].x : pk[i][k].y) >> ((j & 3) * 8)) & 0xFFu;
              if (t == want) g[j] += d[i][k][j];
            }
          }
        }
        store8<T>(dx + (((int64_t)b * H + hi) * W + wi) * C + (cv << 3), g);
      }
    }
  }
}

static bool pool3s2_ok(int B, int H, int W, int C, int k, int stride, int pad_h, int pad_w, int Ho, int Wo) {
  if (pf_tuning().pool3s2 == 0) return false;                              // PF_POOL3S2=0: the generic kernels (A-B / tests)
  return k == 3 && stride == 2 && pad_h >= 0 && pad_h <= 1 && pad_w >= 0 && pad_w <= 1 &&
         (int64_t)B * H * W * (C / 8) < (1ll << 31);
}

extern "C" int pf_maxpool_fwd(const void* x, void* y, void* idx, int dtype, int B, int H, int W, int C, int k, int stride,
                              int pad_h, int pad_w, int Ho, int Wo, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || k < 1 || k > 15 || stride < 1 || Ho <= 0 || Wo <= 0)
    return (int)hipErrorInvalidValue;
  if (!pf_aligned16(x) || !pf_aligned16(y) || (idx && (((uintptr_t)idx) & 7))) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  const int grid = pf_grid_for(total, PF_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (pool3s2_ok(B, H, W, C, k, stride, pad_h, pad_w, Ho, Wo) && (dtype == PF_F32 || dtype == PF_BF16)) {
    if (dtype == PF_F32)
      k_maxpool3s2_fwd<float><<<grid, PF_THREADS, 0, st>>>((const float*)x, (float*)y, (uint8_t*)idx, B, H, W, C, pad_h, pad_w, Ho, Wo);
    else
      k_maxpool3s2_fwd<bf16_t><<<grid, PF_THREADS, 0, st>>>((const bf16_t*)x, (bf16_t*)y, (uint8_t*)idx, B, H, W, C, pad_h, pad_w, Ho, Wo);
    PF_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == PF_F32)
    k_maxpool_fwd<float><<<grid, PF_THREADS, 0, st>>>((const float*)x, (float*)y, (uint8_t*)idx, B, H, W, C, k, stride,
                                                       pad_h, pad_w, Ho, Wo);
  else if (dtype == PF_BF16)
    k_maxpool_fwd<bf16_t><<<grid, PF_THREADS, 0, st>>>((const bf16_t*)x, (bf16_t*)y, (uint8_t*)idx, B, H, W, C, k, stride,
                                                        pad_h, pad_w, Ho, Wo);
  else
    return (int)hipErrorInvalidValue;
  PF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pf_maxpool_bwd(const void* dy, const void* idx, void* dx, int dtype, int B, int H, int W, int C, int k,
                              int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || k < 1 || k > 15 || stride < 1 || Ho <= 0 || Wo <= 0 || idx == nullptr)
    return (int)hipErrorInvalidValue;
  if (!pf_aligned16(dy) || !pf_aligned16(dx) || (((uintptr_t)idx) & 7)) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)B * H * W * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  if (pool3s2_ok(B, H, W, C, k, stride, pad_h, pad_w, Ho, Wo) && (dtype == PF_F32 || dtype == PF_BF16)) {
    const int Ha = (H + pad_h + 1) / 2, Wb = (W + pad_w + 1) / 2;        // 2 x 2 blocks of padded input coordinates
    const int g2 = pf_grid_for((int64_t)B * Ha * Wb * (C / 8), PF_THREADS);
    if (dtype == PF_F32)
      k_maxpool3s2_bwd<float><<<g2, PF_THREADS, 0, st>>>((const float*)dy, (const uint8_t*)idx, (float*)dx, B, H, W, C, pad_h, pad_w, Ho, Wo, Ha, Wb);
    else
      k_maxpool3s2_bwd<bf16_t><<<g2, PF_THREADS, 0, st>>>((const bf16_t*)dy, (const uint8_t*)idx, (bf16_t*)dx, B, H, W, C, pad_h, pad_w, Ho, Wo, Ha, Wb);
    PF_LAUNCH_CHECK();
    return 0;
  }
  const int grid = pf_grid_for(total, PF_THREADS);
  if (dtype == PF_F32)
    k_maxpool_bwd<float><<<grid, PF_THREADS, 0, st>>>((const float*)dy, (const uint8_t*)idx, (float*)dx, B, H, W, C, k,
                                                       stride, pad_h, pad_w, Ho, Wo);
  else if (dtype == PF_BF16)
    k_maxpool_bwd<bf16_t><<<grid, PF_THREADS, 0, st>>>((const bf16_t*)dy, (const uint8_t*)idx, (bf16_t*)dx, B, H, W, C, k,
                                                        stride, pad_h, pad_w, Ho, Wo);
  else
    return (int)hipErrorInvalidValue;
  PF_LAUNCH_CHECK();
  return 0;
}
