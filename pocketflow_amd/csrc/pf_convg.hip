// K12, general form: dense convolutions of ANY shape -- forward, backward-data (any stride) and backward-filter -- and the dense
// layer, in float32 or bf16 storage with float32 accumulation.  This is the path of everything the shape-specialised kernels do
// not take:
//   * the float32 PARITY mode (`--compute_dtype float32`, SURVEY 8(d) C2 "an fp32 parity run"): every convolution of the student
//     and the teacher (utils/external/resnet_model.py:92-103 conv2d_fixed_padding, :552 tf.layers.dense; mobilenet_v1.py
//     slim.conv2d; nets/lenet_at_cifar10.py:34-68) -- the matrix cores have no float32-exact bf16 path, so this mode multiplies
//     on the vector ALUs, one fused multiply-add per term, in a fixed order (deterministic, run-to-run bit-identical);
//   * bf16 layers whose channel counts are not multiples of the MFMA tiles' 64 (3-channel images outside the 7x7 stem,
//     LeNet) and the dense layer 2048 -> 1001 with its two backward products.
// Until round 4 these went to MIOpen / rocBLAS through torch.
//
// One implicit-GEMM kernel, three index maps.  C[M x Nc] = A[M x K] * B[K x Nc]:
//   forward          M = imgs*Ho*Wo   Nc = N (out channels)   K = R*S*C     A = X gathered per tap (zero outside the image)
//                                                                            B = W[n][tap][c]                (KRSC)
//   backward-data    M = imgs*H*W     Nc = C (in channels)    K = R*S*N     A = dY at ((h + pad - r) / stride, ...) when the
//                                                                                division is exact and in range, else zero
//                                                                            B = W[n][tap][c]
//   backward-filter  M = N            Nc = R*S*C              K = imgs*Ho*Wo A = dY^T,  B = X gathered per tap; the pixel range
//                                                                            is cut into splits, fp32 slabs, fixed-order reduce
// 64 x 64 x 16 tiles, 256 threads, 4 x 4 accumulators per thread, both operands staged in LDS as float32 [k][row] so that
// the inner product reads two float4 per k.  HBM access: every loader lane fetches 4 consecutive elements along the operand's
// contiguous index (16 bytes float32 / 8 bytes bf16) whenever the channel counts are multiples of 4, scalars otherwise.
#include "pf_common.h"

#define CG_BM 64
#define CG_BN 64
#define CG_BK 16
#define CG_LD 68                      // padded leading dimension of the LDS tiles (floats): 272-byte rows keep float4 alignment
#define CG_T 256

struct CgArgs {
  const void* P;                      // pixel-side operand of the product: X (forward), dY (backward-data), dY (backward-filter)
  const void* Q;                      // W (forward / backward-data), X (backward-filter)
  void* Out;                          // Y | dX | unused (backward-filter writes `slab`)
  const float* bias;                  // forward: per-output-channel bias or null
  float* slab;                        // backward-filter: [splits][N][R*S*C] float32 partial sums
  int imgs, H, W, C, N, R, S, stride, pad_h, pad_w, Ho, Wo;
  int M, Nc, K;                       // GEMM view
  int k_per_split;                    // backward-filter: pixels per split (multiple of CG_BK)
};

template <typename T> __device__ __forceinline__ float cg_ld(const T* p);
template <> __device__ __forceinline__ float cg_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float cg_ld<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void cg_ld4(const T* p, float* o);
template <> __device__ __forceinline__ void cg_ld4<float>(const float* p, float* o) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void cg_ld4<bf16_t>(const bf16_t* p, float* o) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xFFFF0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xFFFF0000u);
}
template <typename T> __device__ __forceinline__ void cg_st(T* p, float v);
template <> __device__ __forceinline__ void cg_st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void cg_st<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// MODE 0 forward, 1 backward-data, 2 backward-filter.  V4: channel counts are multiples of 4 and the tensors 16-byte aligned
// (vector loads along the contiguous index); otherwise every element is fetched on its own.
template <typename T, int MODE, bool V4>
__global__ __launch_bounds__(CG_T) void k_convg(const CgArgs a) {
  __shared__ __attribute__((aligned(16))) float As[CG_BK][CG_LD];
  __shared__ __attribute__((aligned(16))) float Bs[CG_BK][CG_LD];
  const T* __restrict__ P = reinterpret_cast<const T*>(a.P);
  const T* __restrict__ Q = reinterpret_cast<const T*>(a.Q);
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int tiles_n = (a.Nc + CG_BN - 1) / CG_BN;
  const int tile = blockIdx.x;
  const int split = blockIdx.y;                                  // backward-filter only
  const int m0 = (tile / tiles_n) * CG_BM, n0 = (tile % tiles_n) * CG_BN;
  const int RS = a.R * a.S;

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  int k_beg = 0, k_end = a.K;
  if (MODE == 2 || a.slab != nullptr) {                          // the contraction range of this split
    k_beg = split * a.k_per_split;
    k_end = min(a.K, k_beg + a.k_per_split);
  }

  // ---- loader geometry ---------------------------------------------------------------------------------------------------
  // "k4" lanes own one row (or column) of the tile and 4 consecutive k; "r4" lanes own one k and 4 consecutive rows (columns).
  const int l_row = tid >> 2, l_k4 = (tid & 3) * 4;              // k4: row 0..63, k offset 0,4,8,12
  const int l_k = tid >> 4, l_r4 = (tid & 15) * 4;               // r4: k 0..15, row offset 0..60
  // forward / backward-data: the pixel of this lane's A row, decomposed once
  int p_img = 0, p_y = 0, p_x = 0;
  bool p_ok = false;
  if (MODE != 2) {
    const int m = m0 + l_row;
    p_ok = m < a.M;
    if (p_ok) {
      const int hw = (MODE == 0) ? a.Ho * a.Wo : a.H * a.W;
      const int wd = (MODE == 0) ? a.Wo : a.W;
      p_img = m / hw;
      const int rem = m - p_img * hw;
      p_y = rem / wd;
      p_x = rem - p_y * wd;
    }
  }

  for (int k0 = k_beg; k0 < k_end; k0 += CG_BK) {
    // ---- A tile ------------------------------------------------------------------------------------------------------------
    if (MODE == 0) {
      // A(m, k) = X[img][ho*stride + r - pad_h][wo*stride + s - pad_w][c],  k = (r*S + s)*C + c
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const int k = k0 + l_k4;
      if (p_ok && k < k_end) {
        if (V4) {
          const int tap = k / a.C, c = k - tap * a.C;
          const int r = tap / a.S, s = tap - r * a.S;
          const int h = p_y * a.stride + r - a.pad_h, w = p_x * a.stride + s - a.pad_w;
          if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
            cg_ld4<T>(P + ((int64_t)(p_img * a.H + h) * a.W + w) * a.C + c, v);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int kk = k + i;
            if (kk < k_end) {
              const int tap = kk / a.C, c = kk - tap * a.C;
              const int r = tap / a.S, s = tap - r * a.S;
              const int h = p_y * a.stride + r - a.pad_h, w = p_x * a.stride + s - a.pad_w;
              if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
                v[i] = cg_ld<T>(P + ((int64_t)(p_img * a.H + h) * a.W + w) * a.C + c);
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) As[l_k4 + i][l_row] = v[i];
    } else if (MODE == 1) {
      // A(m, k) = dY[img][(h + pad_h - r) / stride][(w + pad_w - s) / stride][n],  k = (r*S + s)*N + n
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const int k = k0 + l_k4;
      if (p_ok && k < k_end) {
#pragma unroll
        for (int i = 0; i < (V4 ? 1 : 4); ++i) {
          const int kk = k + i;
          if (kk >= k_end) break;
          const int tap = kk / a.N, n = kk - tap * a.N;
          const int r = tap / a.S, s = tap - r * a.S;
          const int th = p_y + a.pad_h - r, tw = p_x + a.pad_w - s;
          if (th >= 0 && tw >= 0) {
            const int ho = th / a.stride, wo = tw / a.stride;
            if (ho * a.stride == th && wo * a.stride == tw && ho < a.Ho && wo < a.Wo) {
              const T* src = P + ((int64_t)(p_img * a.Ho + ho) * a.Wo + wo) * a.N + n;
              if (V4) cg_ld4<T>(src, v); else v[i] = cg_ld<T>(src);
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) As[l_k4 + i][l_row] = v[i];
    } else {
      // A(n, p) = dY[p][n]: contiguous along the ROW index -> r4 lanes
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const int p = k0 + l_k, n = m0 + l_r4;
      if (p < k_end) {
        if (V4 && n + 3 < a.M) {
          cg_ld4<T>(P + (int64_t)p * a.N + n, v);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (n + i < a.M) v[i] = cg_ld<T>(P + (int64_t)p * a.N + n + i);
        }
      }
      *reinterpret_cast<float4*>(&As[l_k][l_r4]) = make_float4(v[0], v[1], v[2], v[3]);
    }
    // ---- B tile ------------------------------------------------------------------------------------------------------------
    if (MODE == 0) {
      // B(k, n) = W[n][k]: contiguous along k -> k4 lanes (their "row" is the output channel)
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const int n = n0 + l_row, k = k0 + l_k4;
      if (n < a.Nc && k < k_end) {
        if (V4) {
          cg_ld4<T>(Q + (int64_t)n * a.K + k, v);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (k + i < k_end) v[i] = cg_ld<T>(Q + (int64_t)n * a.K + k + i);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) Bs[l_k4 + i][l_row] = v[i];
    } else if (MODE == 1) {
      // B(k, c) = W[n][tap][c],  k = tap*N + n: contiguous along c -> r4 lanes
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const int k = k0 + l_k, c = n0 + l_r4;
      if (k < k_end) {
        const int tap = k / a.N, n = k - tap * a.N;
        const T* src = Q + ((int64_t)n * RS + tap) * a.C + c;
        if (V4 && c + 3 < a.Nc) {
          cg_ld4<T>(src, v);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (c + i < a.Nc) v[i] = cg_ld<T>(src + i);
        }
      }
      *reinterpret_cast<float4*>(&Bs[l_k][l_r4]) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      // B(p, j) = X[pix(p, tap(j))][c(j)],  j = tap*C + c: contiguous along c -> r4 lanes
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const int p = k0 + l_k, j = n0 + l_r4;
      if (p < k_end && j < a.Nc) {
        const int hw = a.Ho * a.Wo;
        const int img = p / hw, rem = p - img * hw;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
#pragma unroll
        for (int i = 0; i < (V4 ? 1 : 4); ++i) {
          const int jj = j + i;
          if (jj >= a.Nc) break;
          const int tap = jj / a.C, c = jj - tap * a.C;
          const int r = tap / a.S, s = tap - r * a.S;
          const int h = ho * a.stride + r - a.pad_h, w = wo * a.stride + s - a.pad_w;
          if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) {
            const T* src = Q + ((int64_t)(img * a.H + h) * a.W + w) * a.C + c;
            if (V4) cg_ld4<T>(src, v); else v[i] = cg_ld<T>(src);
          }
        }
      }
      *reinterpret_cast<float4*>(&Bs[l_k][l_r4]) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    // ---- 16 rank-1 updates of the 4 x 4 block, k ascending: one fused multiply-add per term ------------------------------------
#pragma unroll
    for (int kk = 0; kk < CG_BK; ++kk) {
      const float4 av = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
      const float ar[4] = {av.x, av.y, av.z, av.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(ar[i], br[j], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue --------------------------------------------------------------------------------------------------------------
  if (MODE == 2 || a.slab != nullptr) {
    float* out = a.slab + (int64_t)split * a.M * a.Nc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + ty * 4 + i;
      if (m >= a.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + tx * 4 + j;
        if (n < a.Nc) out[(int64_t)m * a.Nc + n] = acc[i][j];
      }
    }
  } else {
    T* out = reinterpret_cast<T*>(a.Out);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + ty * 4 + i;
      if (m >= a.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + tx * 4 + j;
        if (n < a.Nc) {
          float v = acc[i][j];
          if (MODE == 0 && a.bias != nullptr) v = v + a.bias[n];
          cg_st<T>(out + (int64_t)m * a.Nc + n, v);
        }
      }
    }
  }
}

// out[i] = sum over splits (ascending) of slab[s][i] (+ bias[i % Nc])
template <typename T>
__global__ __launch_bounds__(CG_T) void k_convg_reduce(const float* __restrict__ slab, int splits, int64_t n, T* __restrict__ out,
                                                       const float* __restrict__ bias, int Nc) {
  for (int64_t i = (int64_t)blockIdx.x * CG_T + threadIdx.x; i < n; i += (int64_t)gridDim.x * CG_T) {
    float s = slab[i];
    for (int k = 1; k < splits; ++k) s = s + slab[(int64_t)k * n + i];
    if (bias != nullptr) s = s + bias[(int)(i % Nc)];
    cg_st<T>(out + i, s);
  }
}

// Contraction splits of a forward / backward-data launch whose output has too few tiles to fill the chip (the dense layer: 256 x 1001
// outputs = 64 tiles, 2048 terms each): 0 = one launch writes the output, > 0 = that many float32 slabs + a fixed-order reduction.
static int cg_small_splits(int M, int Nc, int K) {
  // a function of the SHAPE only (round 5): the summation order of a layer must not depend on how large a workspace the
  // caller happens to hold -- a workspace that is too small for the split the shape asks for is an error, not a different sum
  const int64_t tiles = (int64_t)((M + CG_BM - 1) / CG_BM) * ((Nc + CG_BN - 1) / CG_BN);
  if (tiles >= 256 || K < 512) return 0;
  int s = (int)((512 + tiles - 1) / tiles);
  const int max_s = K / 256;
  if (s > max_s) s = max_s;
  if (s > 16) s = 16;
  if (s < 2 || (int64_t)s * M * Nc > ((int64_t)1 << 24)) return 0;      // (64 MiB of float32 slabs at most)
  return s;
}

template <typename T, int MODE>
static int cg_launch_split(CgArgs& a, bool v4, int dtype_unused, float* slab, int64_t slab_elems, hipStream_t st) {
  (void)dtype_unused;
  const int tiles = ((a.M + CG_BM - 1) / CG_BM) * ((a.Nc + CG_BN - 1) / CG_BN);
  const int splits = (slab != nullptr) ? cg_small_splits(a.M, a.Nc, a.K) : 0;
  if (splits > 0 && (int64_t)splits * a.M * a.Nc > slab_elems) return (int)hipErrorInvalidValue;
  if (splits == 0) {
    a.slab = nullptr;
    if (v4) k_convg<T, MODE, true><<<dim3((unsigned)tiles, 1, 1), CG_T, 0, st>>>(a);
    else k_convg<T, MODE, false><<<dim3((unsigned)tiles, 1, 1), CG_T, 0, st>>>(a);
    PF_LAUNCH_CHECK();
    return 0;
  }
  int kps = (a.K + splits - 1) / splits;
  kps = ((kps + CG_BK - 1) / CG_BK) * CG_BK;
  a.k_per_split = kps;
  a.slab = slab;
  const int used = (a.K + kps - 1) / kps;
  const float* bias = a.bias;
  a.bias = nullptr;                                            // added once, by the reduction
  if (v4) k_convg<T, MODE, true><<<dim3((unsigned)tiles, (unsigned)used, 1), CG_T, 0, st>>>(a);
  else k_convg<T, MODE, false><<<dim3((unsigned)tiles, (unsigned)used, 1), CG_T, 0, st>>>(a);
  PF_LAUNCH_CHECK();
  const int64_t n = (int64_t)a.M * a.Nc;
  k_convg_reduce<T><<<pf_grid_for(n, CG_T), CG_T, 0, st>>>(slab, used, n, reinterpret_cast<T*>(a.Out), bias, a.Nc);
  PF_LAUNCH_CHECK();
  return 0;
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static bool cg_geom_ok(int imgs, int H, int W, int C, int N, int R, int S, int stride, int pad_h, int pad_w, int Ho, int Wo) {
  if (imgs <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad_h < 0 || pad_w < 0 || Ho <= 0 || Wo <= 0)
    return false;
  // every output position must lie inside the (padded) input's reach: (Ho-1)*stride - pad + (R-1) may exceed H-1 only by the pad
  if ((int64_t)imgs * H * W * C >= ((int64_t)1 << 40) || (int64_t)imgs * Ho * Wo >= ((int64_t)1 << 31) ||
      (int64_t)imgs * H * W >= ((int64_t)1 << 31) || (int64_t)R * S * C >= ((int64_t)1 << 31) || (int64_t)R * S * N >= ((int64_t)1 << 31))
    return false;
  return true;
}

template <typename T, int MODE>
static int cg_launch(const CgArgs& a, bool v4, dim3 grid, hipStream_t st) {
  if (v4) k_convg<T, MODE, true><<<grid, CG_T, 0, st>>>(a);
  else k_convg<T, MODE, false><<<grid, CG_T, 0, st>>>(a);
  PF_LAUNCH_CHECK();
  return 0;
}

// slab / slab_elems: optional float32 workspace (null / 0: none: one ascending sum per output).  With one, a launch whose output is too
// small to fill the chip splits its contraction into pf_convg_small_splits(M, Nc, K) slabs -- a function of the shape alone, so that a
// layer sums in the same order whatever else has grown the caller's workspace; a workspace smaller than splits * M * Nc is an error.
extern "C" int pf_convg_small_splits(int M, int Nc, int K) {
  if (M <= 0 || Nc <= 0 || K <= 0) return 0;
  return cg_small_splits(M, Nc, K);
}

extern "C" int pf_convg_fwd(const void* x, const void* w, const float* bias, void* y, int dtype, int imgs, int H, int W, int C,
                            int N, int R, int S, int stride, int pad_h, int pad_w, int Ho, int Wo, float* slab, int64_t slab_elems,
                            void* stream) {
  if (!cg_geom_ok(imgs, H, W, C, N, R, S, stride, pad_h, pad_w, Ho, Wo)) return (int)hipErrorInvalidValue;
  CgArgs a{};
  a.P = x; a.Q = w; a.Out = y; a.bias = bias;
  a.imgs = imgs; a.H = H; a.W = W; a.C = C; a.N = N; a.R = R; a.S = S; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.Ho = Ho; a.Wo = Wo;
  a.M = imgs * Ho * Wo; a.Nc = N; a.K = R * S * C;
  const bool v4 = (C % 4 == 0) && pf_aligned16(x) && pf_aligned16(w);
  if (dtype == PF_F32) return cg_launch_split<float, 0>(a, v4, dtype, slab, slab_elems, (hipStream_t)stream);
  if (dtype == PF_BF16) return cg_launch_split<bf16_t, 0>(a, v4, dtype, slab, slab_elems, (hipStream_t)stream);
  return (int)hipErrorInvalidValue;
}

extern "C" int pf_convg_bwd_data(const void* dy, const void* w, void* dx, int dtype, int imgs, int H, int W, int C, int N, int R,
                                 int S, int stride, int pad_h, int pad_w, int Ho, int Wo, float* slab, int64_t slab_elems, void* stream) {
  if (!cg_geom_ok(imgs, H, W, C, N, R, S, stride, pad_h, pad_w, Ho, Wo)) return (int)hipErrorInvalidValue;
  CgArgs a{};
  a.P = dy; a.Q = w; a.Out = dx;
  a.imgs = imgs; a.H = H; a.W = W; a.C = C; a.N = N; a.R = R; a.S = S; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.Ho = Ho; a.Wo = Wo;
  a.M = imgs * H * W; a.Nc = C; a.K = R * S * N;
  const bool v4 = (C % 4 == 0) && (N % 4 == 0) && pf_aligned16(dy) && pf_aligned16(w);
  if (dtype == PF_F32) return cg_launch_split<float, 1>(a, v4, dtype, slab, slab_elems, (hipStream_t)stream);
  if (dtype == PF_BF16) return cg_launch_split<bf16_t, 1>(a, v4, dtype, slab, slab_elems, (hipStream_t)stream);
  return (int)hipErrorInvalidValue;
}

// pixel splits of the backward-filter product: enough workgroups to fill the chip, at least 256 pixels per split
extern "C" int pf_convg_wrw_splits(int imgs, int C, int N, int R, int S, int Ho, int Wo) {
  const int64_t pixels = (int64_t)imgs * Ho * Wo;
  const int64_t tiles = (int64_t)((N + CG_BM - 1) / CG_BM) * ((R * S * C + CG_BN - 1) / CG_BN);
  int64_t s = (1024 + tiles - 1) / tiles;
  const int64_t max_s = (pixels + 255) / 256;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 4096) s = 4096;
  return (int)s;
}

// dw [N][R][S][C] (dw_dtype: float32 or bf16);  slab: float32 workspace of pf_convg_wrw_splits(...) * N * R*S*C elements
extern "C" int pf_convg_wrw(const void* dy, const void* x, void* dw, int dtype, int dw_dtype, float* slab, int imgs, int H, int W,
                            int C, int N, int R, int S, int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream) {
  if (!cg_geom_ok(imgs, H, W, C, N, R, S, stride, pad_h, pad_w, Ho, Wo) || slab == nullptr) return (int)hipErrorInvalidValue;
  CgArgs a{};
  a.P = dy; a.Q = x; a.slab = slab;
  a.imgs = imgs; a.H = H; a.W = W; a.C = C; a.N = N; a.R = R; a.S = S; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.Ho = Ho; a.Wo = Wo;
  a.M = N; a.Nc = R * S * C; a.K = imgs * Ho * Wo;
  const int splits = pf_convg_wrw_splits(imgs, C, N, R, S, Ho, Wo);
  int kps = (a.K + splits - 1) / splits;
  kps = ((kps + CG_BK - 1) / CG_BK) * CG_BK;
  a.k_per_split = kps;
  const int used = (a.K + kps - 1) / kps;                   // <= splits; the slabs of unused splits are never read
  const bool v4 = (C % 4 == 0) && (N % 4 == 0) && pf_aligned16(dy) && pf_aligned16(x);
  const dim3 grid((unsigned)(((a.M + CG_BM - 1) / CG_BM) * ((a.Nc + CG_BN - 1) / CG_BN)), (unsigned)used, 1);
  int rc;
  if (dtype == PF_F32) rc = cg_launch<float, 2>(a, v4, grid, (hipStream_t)stream);
  else if (dtype == PF_BF16) rc = cg_launch<bf16_t, 2>(a, v4, grid, (hipStream_t)stream);
  else return (int)hipErrorInvalidValue;
  if (rc != 0) return rc;
  const int64_t n = (int64_t)a.M * a.Nc;
  const int rgrid = pf_grid_for(n, CG_T);
  if (dw_dtype == PF_F32) k_convg_reduce<float><<<rgrid, CG_T, 0, (hipStream_t)stream>>>(slab, used, n, (float*)dw, nullptr, 1);
  else if (dw_dtype == PF_BF16) k_convg_reduce<bf16_t><<<rgrid, CG_T, 0, (hipStream_t)stream>>>(slab, used, n, (bf16_t*)dw, nullptr, 1);
  else return (int)hipErrorInvalidValue;
  PF_LAUNCH_CHECK();
  return 0;
}
