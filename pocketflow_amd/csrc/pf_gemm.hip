// K12: bf16 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x32_bf16), fp32 accumulate.
// In NHWC a 1x1 stride-1 convolution is a row-major GEMM over [rows = N*H*W][C]:
//   forward    Y[rows][Cout]  = X[rows][Cin]   * W[Cout][Cin]^T      -> pf_gemm_bf16_nt
//   bwd-data   dX[rows][Cin]  = dY[rows][Cout] * W[Cout][Cin]        -> pf_gemm_bf16_nn
//   bwd-filter dW[Cout][Cin] += dY[rows][Cout]^T * X[rows][Cin]      -> pf_gemm_bf16_tn (split-K)
//
// Tiling (64-lane wavefronts, not warp-shaped): 128x128 output tile per 256-thread workgroup,
// 2x2 wavefronts, each wavefront owns a 64x64 sub-tile = 4x4 MFMA 16x16 accumulators (64 fp32
// VGPR/lane).  K step 64: both operands are staged in LDS as [row][k] with a +8-element pad
// (row stride 144 B => the 16-byte fragment reads of 16 consecutive rows fall on distinct 16-byte
// bank slots), the next K tile is prefetched into registers while the current one is consumed.
// K-major operands (nn / tn) are transposed while being staged.  Workgroup ids are remapped so
// that consecutive tiles of one XCD (id % 8) share A panels in that XCD's L2.
//
// Reference ops replaced (paths under /root/reference): utils/external/resnet_model.py:92-103
// (tf.layers.conv2d), learners/uniform_quantization/utils.py:92-103 (tf.nn.conv2d / tf.matmul on
// the quantised kernel) and their TF gradients Conv2DBackpropInput / Conv2DBackpropFilter.
#include "pf_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define GM_BM 128
#define GM_BN 128
#define GM_BK 64
#define GM_LDK (GM_BK + 8)

enum { GM_NT = 0, GM_NN = 1, GM_TN = 2 };

// ---- staging helpers -------------------------------------------------------------------------
// K-contiguous operand: tile rows [r0, r0+128) x k [k0, k0+64) of P[rows][ld=K]
__device__ __forceinline__ void g_load_kcontig(const bf16_t* __restrict__ P, int rows, int ld, int r0,
                                               int k0, int kend, uint4* regs) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = threadIdx.x + i * PF_THREADS;
    const int row = p >> 3, kp = p & 7;
    const int r = r0 + row, k = k0 + kp * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < rows && k < kend) v = *reinterpret_cast<const uint4*>(P + (int64_t)r * ld + k);
    regs[i] = v;
  }
}
__device__ __forceinline__ void s_store_kcontig(bf16_t* __restrict__ S, const uint4* regs) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = threadIdx.x + i * PF_THREADS;
    const int row = p >> 3, kp = p & 7;
    *reinterpret_cast<uint4*>(S + row * GM_LDK + kp * 8) = regs[i];
  }
}
// K-major operand: tile k [k0, k0+64) x cols [c0, c0+128) of P[K][ld=cols]
__device__ __forceinline__ void g_load_kmajor(const bf16_t* __restrict__ P, int cols, int ld, int c0,
                                              int k0, int kend, uint4* regs) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = threadIdx.x + i * PF_THREADS;
    const int krow = p >> 4, seg = p & 15;
    const int k = k0 + krow, c = c0 + seg * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (k < kend && c < cols) v = *reinterpret_cast<const uint4*>(P + (int64_t)k * ld + c);
    regs[i] = v;
  }
}
__device__ __forceinline__ void s_store_kmajor(bf16_t* __restrict__ S, const uint4* regs) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = threadIdx.x + i * PF_THREADS;
    const int krow = p >> 4, seg = p & 15;
    const uint32_t w[4] = {regs[i].x, regs[i].y, regs[i].z, regs[i].w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bf16_t e = (bf16_t)((w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
      S[(seg * 8 + j) * GM_LDK + krow] = e;
    }
  }
}

template <int MODE, typename TC, bool ATOMIC>
__global__ __launch_bounds__(PF_THREADS) void k_gemm_bf16(const bf16_t* __restrict__ A,
                                                          const bf16_t* __restrict__ B, TC* __restrict__ C,
                                                          int M, int N, int K, int kchunk, int tiles_m,
                                                          int tiles_n) {
  __shared__ __attribute__((aligned(16))) bf16_t As[GM_BM * GM_LDK];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[GM_BN * GM_LDK];

  // XCD-aware tile order: hardware places workgroup b on XCD b % 8; give each XCD a contiguous
  // run of tiles (bijective remap, also when the tile count is not a multiple of 8)
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int xcd = wg & 7, idx = wg >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // column-of-tiles fastest => neighbouring workgroups share the A panel (rows), W stays in L2
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * GM_BM, n0 = tn * GM_BN;
  const int kbeg = blockIdx.y * kchunk;
  const int kend = (kbeg + kchunk < K) ? (kbeg + kchunk) : K;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fk = (lane >> 4) * 8;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  auto gload = [&](int k0) {
    if (MODE == GM_TN) g_load_kmajor(A, M, M, m0, k0, kend, ra);
    else g_load_kcontig(A, M, K, m0, k0, kend, ra);
    if (MODE == GM_NT) g_load_kcontig(B, N, K, n0, k0, kend, rb);
    else g_load_kmajor(B, N, N, n0, k0, kend, rb);
  };
  auto sstore = [&]() {
    if (MODE == GM_TN) s_store_kmajor(As, ra); else s_store_kcontig(As, ra);
    if (MODE == GM_NT) s_store_kcontig(Bs, rb); else s_store_kmajor(Bs, rb);
  };

  if (kbeg < kend) gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += GM_BK) {
    sstore();
    __syncthreads();
    if (k0 + GM_BK < kend) gload(k0 + GM_BK);      // prefetch next tile into registers
#pragma unroll
    for (int kk = 0; kk < GM_BK / 32; ++kk) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(As + (wm * 64 + i * 16 + frow) * GM_LDK + kk * 32 + fk);
        bfr[i] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 64 + i * 16 + frow) * GM_LDK + kk * 32 + fk);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
  const int crow = (lane >> 4) * 4, ccol = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + ccol;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + i * 16 + crow + r;
        if (row < M) {
          TC* dst = C + (int64_t)row * N + col;
          if (ATOMIC) atomicAdd(reinterpret_cast<float*>(dst), acc[i][j][r]);
          else store_one<TC>(dst, acc[i][j][r]);
        }
      }
    }
}

template <int MODE>
static int launch_gemm(const void* A, const void* B, void* C, int M, int N, int K, int out_dtype,
                       hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
  if (!pf_aligned16(A) || !pf_aligned16(B)) return (int)hipErrorInvalidValue;
  if (MODE != GM_TN && (K % 8)) return (int)hipErrorInvalidValue;
  if (MODE == GM_NN && (N % 8)) return (int)hipErrorInvalidValue;
  if (MODE == GM_TN && ((M % 8) || (N % 8))) return (int)hipErrorInvalidValue;
  const int tiles_m = (M + GM_BM - 1) / GM_BM, tiles_n = (N + GM_BN - 1) / GM_BN;
  const bf16_t* a = (const bf16_t*)A;
  const bf16_t* b = (const bf16_t*)B;
  if (MODE == GM_TN) {
    if (out_dtype != PF_F32) return (int)hipErrorInvalidValue;
    // split K (= rows, up to ~10^6) so that the grid fills 256 CUs several times over
    int splits = (2048 + tiles_m * tiles_n - 1) / (tiles_m * tiles_n);
    int kchunk = (K + splits - 1) / splits;
    kchunk = ((kchunk + GM_BK - 1) / GM_BK) * GM_BK;
    if (kchunk < GM_BK * 4) kchunk = GM_BK * 4;
    splits = (K + kchunk - 1) / kchunk;
    dim3 grid(tiles_m * tiles_n, splits);
    k_gemm_bf16<GM_TN, float, true><<<grid, PF_THREADS, 0, st>>>(a, b, (float*)C, M, N, K, kchunk, tiles_m, tiles_n);
  } else {
    dim3 grid(tiles_m * tiles_n, 1);
    if (out_dtype == PF_BF16)
      k_gemm_bf16<MODE, bf16_t, false><<<grid, PF_THREADS, 0, st>>>(a, b, (bf16_t*)C, M, N, K, K, tiles_m, tiles_n);
    else if (out_dtype == PF_F32)
      k_gemm_bf16<MODE, float, false><<<grid, PF_THREADS, 0, st>>>(a, b, (float*)C, M, N, K, K, tiles_m, tiles_n);
    else return (int)hipErrorInvalidValue;
  }
  PF_LAUNCH_CHECK();
  return 0;
}

extern "C" int pf_gemm_bf16_nt(const void* A, const void* B, void* C, int M, int N, int K, int out_dtype,
                               void* stream) {
  return launch_gemm<GM_NT>(A, B, C, M, N, K, out_dtype, (hipStream_t)stream);
}
extern "C" int pf_gemm_bf16_nn(const void* A, const void* B, void* C, int M, int N, int K, int out_dtype,
                               void* stream) {
  return launch_gemm<GM_NN>(A, B, C, M, N, K, out_dtype, (hipStream_t)stream);
}
extern "C" int pf_gemm_bf16_tn(const void* A, const void* B, void* C, int M, int N, int K, int out_dtype,
                               void* stream) {
  return launch_gemm<GM_TN>(A, B, C, M, N, K, out_dtype, (hipStream_t)stream);
}
