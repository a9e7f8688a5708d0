// K12 for MobileNet-v1: the depthwise 3x3 convolutions (DepthwiseConv2dNative, utils/external/mobilenet_v1.py:264-292 =
// slim.separable_conv2d(num_outputs=None)), forward, backward-data and backward-filter, NHWC activations, kernels stored
// [C][kh][kw] (graph.Variable 'depthwise': CRS).  The reference pads 'SAME' the TensorFlow way: total = max((out-1)*stride
// + k - in, 0), floor(total / 2) in FRONT and the rest behind -- asymmetric for stride 2 on even sizes (0 in front, 1 behind),
// so the kernels take the front pads and bound-check both sides.
//
//   y [b][ho][wo][c] = sum_{r,s} x[b][ho*st + r - ph][wo*st + s - pw][c] * w[c][r][s]
//   dx[b][hi][wi][c] = sum_{r,s : (hi + ph - r) % st == 0, (wi + pw - s) % st == 0} dy[b][(hi+ph-r)/st][(wi+pw-s)/st][c] * w[c][r][s]
//   dw[c][r][s]      = sum_{b,ho,wo} dy[b][ho][wo][c] * x[b][ho*st + r - ph][wo*st + s - pw][c]
//
// MI355X mapping.  One multiply-add per loaded element: these are HBM-bound layers (1 read + 1 write of the activation per
// pass; MobileNet-v1 at batch 256 moves 5.0 M elements per image through them), there is nothing for the matrix cores.
// A lane owns 8 consecutive channels (one 16-byte vector) and walks STRIPS of 4 output pixels of one row, so that
// consecutive lanes read consecutive 16-byte groups of one pixel (coalesced 64 B .. 2 KiB runs) and the 3 x (3 + 3*st)
// input vectors of a strip are loaded once for its 4 outputs; the vertical re-use between rows and the horizontal halo
// between strips are served by the L1 / L2 (the input tensor is read once from HBM).  The 72 kernel taps of the lane's
// channels live in registers for the whole launch.  The forward kernel leaves the per-channel {sum, sum of squares, min,
// max} of the values it stored (the statistics the BatchNorm behind every depthwise layer needs, mobilenet_v1.py:279-292)
// in partial[G][4][C], reduced in a fixed order: one pass over the tensor less per layer, deterministic.  Backward-filter
// keeps 9 x 8 float32 accumulators per lane, combines the lanes of a workgroup through LDS and writes one [C][9] slab per
// workgroup; a second launch adds the slabs in a fixed order (bit-reproducible, like pf_wrw_reduce).
#include "pf_common.h"

#define DW_T 256
#define DW_PW 4

// Round 5: every activation load of these kernels goes through a buffer descriptor with a per-lane BYTE offset, and a position
// outside the image gets an offset outside the buffer -- the hardware returns zeros, no branch.  Rounds 3-4 wrote
// `if (inside) load8(...) else zeros` per vector: hipcc turns that into a branch around every load with `s_waitcnt vmcnt(0)` behind
// it (cdna_hip_programming.md, "per-element register-or-load select"), i.e. 18 DEPENDENT round trips to L2 / HBM per strip of four
// output pixels -- the kernels ran at 2.5-2.9x their HBM floor (profiles/r05_depthwise_layers_before.txt), latency-bound at two
// wavefronts per SIMD.  Now the 18 (+ 4) loads of a strip are issued back to back and waited for once.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t dw_rsrc_t;
#define DW_MAKE_RSRC(p, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(p), (short)0, (int)(bytes), 0x00020000)
typedef uint32_t dw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 dw_load16(dw_rsrc_t rs, uint32_t off) {
  const dw_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
#else
typedef int dw_rsrc_t;
#define DW_MAKE_RSRC(p, bytes) 0
__device__ __forceinline__ uint4 dw_load16(dw_rsrc_t, uint32_t) { return make_uint4(0, 0, 0, 0); }
#endif
#define DW_OOB 0x80000000u

// eight consecutive channels of one pixel as they sit in memory: one 16-byte vector (bf16) or two (float32)
template <typename T> struct DwRaw;
template <> struct DwRaw<bf16_t> { uint4 a; };
template <> struct DwRaw<float> { uint4 a, b; };
template <typename T> __device__ __forceinline__ void dw_fetch(dw_rsrc_t rs, uint32_t off, DwRaw<T>& r);
template <> __device__ __forceinline__ void dw_fetch<bf16_t>(dw_rsrc_t rs, uint32_t off, DwRaw<bf16_t>& r) { r.a = dw_load16(rs, off); }
template <> __device__ __forceinline__ void dw_fetch<float>(dw_rsrc_t rs, uint32_t off, DwRaw<float>& r) {
  r.a = dw_load16(rs, off);
  r.b = dw_load16(rs, off == DW_OOB ? DW_OOB : off + 16u);
}
template <typename T> __device__ __forceinline__ void dw_unpack(const DwRaw<T>& r, float* o);
template <> __device__ __forceinline__ void dw_unpack<bf16_t>(const DwRaw<bf16_t>& r, float* o) {
  o[0] = __uint_as_float(r.a.x << 16); o[1] = __uint_as_float(r.a.x & 0xFFFF0000u);
  o[2] = __uint_as_float(r.a.y << 16); o[3] = __uint_as_float(r.a.y & 0xFFFF0000u);
  o[4] = __uint_as_float(r.a.z << 16); o[5] = __uint_as_float(r.a.z & 0xFFFF0000u);
  o[6] = __uint_as_float(r.a.w << 16); o[7] = __uint_as_float(r.a.w & 0xFFFF0000u);
}
template <> __device__ __forceinline__ void dw_unpack<float>(const DwRaw<float>& r, float* o) {
  o[0] = __uint_as_float(r.a.x); o[1] = __uint_as_float(r.a.y); o[2] = __uint_as_float(r.a.z); o[3] = __uint_as_float(r.a.w);
  o[4] = __uint_as_float(r.b.x); o[5] = __uint_as_float(r.b.y); o[6] = __uint_as_float(r.b.z); o[7] = __uint_as_float(r.b.w);
}

struct DwArgs {
  const void* x;       // fwd: input; bwd-data: dy; wrw: x
  const void* w;       // [C][3][3] in the activation dtype, always the forward kernel (`flip` reverses the tap order)
  void* y;             // fwd: output; bwd-data: dx
  const void* dy;      // wrw only
  float* partial;      // fwd: [G][4][C] statistics or null; wrw: [G][C][9] slabs
  int B, H, W, C, Ho, Wo, stride, ph, pw;
  int flip;            // forward kernel used as stride-1 backward-data: taps read in reverse order
  int strips_w;        // strips per output row
  int64_t total;       // strips in the tensor
  uint32_t x_bytes, dy_bytes;   // sizes of the tensors behind `x` / `dy` (buffer descriptors; < 2^31)
};

template <typename T> __device__ __forceinline__ float dw_round(float v);
template <> __device__ __forceinline__ float dw_round<float>(float v) { return v; }
template <> __device__ __forceinline__ float dw_round<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// forward (and stride-1 backward-data, which is the same sum over the flipped kernel with front pads k-1-p)
template <typename T, int ST>
__global__ __launch_bounds__(DW_T) void k_dw_fwd(const DwArgs a) {
  __shared__ float red[4 * DW_T * 8];
  const int C8 = a.C >> 3, SPB = DW_T / C8;
  const int cg = threadIdx.x % C8, sl = threadIdx.x / C8;
  const dw_rsrc_t rsx = DW_MAKE_RSRC(a.x, a.x_bytes);
  const T* __restrict__ w = (const T*)a.w;
  T* __restrict__ y = (T*)a.y;
  float wr[9][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t][j] = load_one<T>(w + (int64_t)(cg * 8 + j) * 9 + (a.flip ? 8 - t : t));
  float st_s[8], st_q[8], st_mn[8], st_mx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; st_mn[j] = INFINITY; st_mx[j] = -INFINITY; }
  constexpr int NV = (DW_PW - 1) * ST + 3;
  for (int64_t strip = (int64_t)blockIdx.x * SPB + sl; strip < a.total; strip += (int64_t)gridDim.x * SPB) {
    const int sw = (int)(strip % a.strips_w);
    const int64_t t = strip / a.strips_w;
    const int ho = (int)(t % a.Ho), b = (int)(t / a.Ho);
    const int wo0 = sw * DW_PW;
    float acc[DW_PW][8];
#pragma unroll
    for (int p = 0; p < DW_PW; ++p)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[p][j] = 0.f;
    // all 3 x NV vectors of the strip are requested before the first one is used (positions outside the image: zeros)
    DwRaw<T> raw[3][NV];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * ST + r - a.ph;
      const bool rok = (unsigned)hi < (unsigned)a.H;
      const uint32_t rbase = (uint32_t)((((int64_t)(b * a.H + hi) * a.W) * a.C + cg * 8) * (int)sizeof(T));
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int wi = wo0 * ST - a.pw + i;
        const bool ok = rok && (unsigned)wi < (unsigned)a.W;
        dw_fetch<T>(rsx, ok ? rbase + (uint32_t)(wi * a.C * (int)sizeof(T)) : DW_OOB, raw[r][i]);
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float v[NV][8];
#pragma unroll
      for (int i = 0; i < NV; ++i) dw_unpack<T>(raw[r][i], v[i]);
#pragma unroll
      for (int p = 0; p < DW_PW; ++p)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[p][j] = fmaf(v[p * ST + s][j], wr[r * 3 + s][j], acc[p][j]);
    }
#pragma unroll
    for (int p = 0; p < DW_PW; ++p) {
      if (wo0 + p >= a.Wo) continue;
      store8<T>(y + ((int64_t)(b * a.Ho + ho) * a.Wo + wo0 + p) * a.C + cg * 8, acc[p]);
      if (a.partial != nullptr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = dw_round<T>(acc[p][j]);                 // statistics see the stored values
          st_s[j] += f;
          st_q[j] = fmaf(f, f, st_q[j]);
          st_mn[j] = pf_acc_min(st_mn[j], f);
          st_mx[j] = pf_acc_max(st_mx[j], f);
        }
      }
    }
  }
  if (a.partial != nullptr) {
    // lanes with equal channel group (sl = 0 .. SPB-1) -> one row of partial, fixed order
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(0 * SPB + sl) * a.C + cg * 8 + j] = st_s[j];
      red[(1 * SPB + sl) * a.C + cg * 8 + j] = st_q[j];
      red[(2 * SPB + sl) * a.C + cg * 8 + j] = st_mn[j];
      red[(3 * SPB + sl) * a.C + cg * 8 + j] = st_mx[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * a.C; i += DW_T) {
      const int stat = i / a.C, c = i - stat * a.C;
      float v = red[(stat * SPB) * a.C + c];
      for (int k = 1; k < SPB; ++k) {
        const float u = red[(stat * SPB + k) * a.C + c];
        v = (stat < 2) ? (v + u) : (stat == 2 ? fminf(v, u) : fmaxf(v, u));
      }
      a.partial[((int64_t)blockIdx.x * 4 + stat) * a.C + c] = v;
    }
  }
}

// backward-data for any stride (gather form): one lane = 8 channels of one INPUT pixel
template <typename T>
__global__ __launch_bounds__(DW_T) void k_dw_bwd_data(const DwArgs a) {
  const int C8 = a.C >> 3, SPB = DW_T / C8;
  const int cg = threadIdx.x % C8, sl = threadIdx.x / C8;
  const dw_rsrc_t rsy = DW_MAKE_RSRC(a.x, a.x_bytes);
  const T* __restrict__ w = (const T*)a.w;
  T* __restrict__ dx = (T*)a.y;
  float wr[9][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t][j] = load_one<T>(w + (int64_t)(cg * 8 + j) * 9 + t);
  const int64_t npix = (int64_t)a.B * a.H * a.W;
  for (int64_t pix = (int64_t)blockIdx.x * SPB + sl; pix < npix; pix += (int64_t)gridDim.x * SPB) {
    const int wi = (int)(pix % a.W);
    const int64_t t = pix / a.W;
    const int hi = (int)(t % a.H), b = (int)(t / a.H);
    // the nine candidate taps, requested at once: a tap that does not hit an output pixel reads zeros
    DwRaw<T> raw[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int th = hi + a.ph - r;
      const int ho = th / a.stride;
      const bool rok = th >= 0 && (th - ho * a.stride) == 0 && ho < a.Ho;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int tw = wi + a.pw - s;
        const int wo = tw / a.stride;
        const bool ok = rok && tw >= 0 && (tw - wo * a.stride) == 0 && wo < a.Wo;
        dw_fetch<T>(rsy, ok ? (uint32_t)((((int64_t)(b * a.Ho + ho) * a.Wo + wo) * a.C + cg * 8) * (int)sizeof(T)) : DW_OOB, raw[r * 3 + s]);
      }
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      float g[8];
      dw_unpack<T>(raw[t9], g);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(g[j], wr[t9][j], acc[j]);
    }
    store8<T>(dx + pix * a.C + cg * 8, acc);
  }
}

// backward-filter: strips of 4 output pixels as in the forward kernel; slab [G][C][9]
template <typename T, int ST>
__global__ __launch_bounds__(DW_T) void k_dw_wrw(const DwArgs a) {
  __shared__ float red[DW_T * 8];                       // one tap at a time: [SPB][C]
  const int C8 = a.C >> 3, SPB = DW_T / C8;
  const int cg = threadIdx.x % C8, sl = threadIdx.x / C8;
  const dw_rsrc_t rsx = DW_MAKE_RSRC(a.x, a.x_bytes);
  const dw_rsrc_t rsy = DW_MAKE_RSRC(a.dy, a.dy_bytes);
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  constexpr int NV = (DW_PW - 1) * ST + 3;
  for (int64_t strip = (int64_t)blockIdx.x * SPB + sl; strip < a.total; strip += (int64_t)gridDim.x * SPB) {
    const int sw = (int)(strip % a.strips_w);
    const int64_t t = strip / a.strips_w;
    const int ho = (int)(t % a.Ho), b = (int)(t / a.Ho);
    const int wo0 = sw * DW_PW;
    // the 4 gradient vectors and the 3 x NV input vectors of the strip, requested at once
    DwRaw<T> graw[DW_PW], raw[3][NV];
#pragma unroll
    for (int p = 0; p < DW_PW; ++p)
      dw_fetch<T>(rsy, (wo0 + p < a.Wo) ? (uint32_t)((((int64_t)(b * a.Ho + ho) * a.Wo + wo0 + p) * a.C + cg * 8) * (int)sizeof(T)) : DW_OOB, graw[p]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * ST + r - a.ph;
      const bool rok = (unsigned)hi < (unsigned)a.H;
      const uint32_t rbase = (uint32_t)((((int64_t)(b * a.H + hi) * a.W) * a.C + cg * 8) * (int)sizeof(T));
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int wi = wo0 * ST - a.pw + i;
        const bool ok = rok && (unsigned)wi < (unsigned)a.W;
        dw_fetch<T>(rsx, ok ? rbase + (uint32_t)(wi * a.C * (int)sizeof(T)) : DW_OOB, raw[r][i]);
      }
    }
    float g[DW_PW][8];
#pragma unroll
    for (int p = 0; p < DW_PW; ++p) dw_unpack<T>(graw[p], g[p]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float v[NV][8];
#pragma unroll
      for (int i = 0; i < NV; ++i) dw_unpack<T>(raw[r][i], v[i]);
#pragma unroll
      for (int p = 0; p < DW_PW; ++p)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[r * 3 + s][j] = fmaf(g[p][j], v[p * ST + s][j], acc[r * 3 + s][j]);
    }
  }
  float* slab = a.partial + (int64_t)blockIdx.x * a.C * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[sl * a.C + cg * 8 + j] = acc[t][j];
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += DW_T) {
      float v = red[c];
      for (int k = 1; k < SPB; ++k) v += red[k * a.C + c];
      slab[(int64_t)c * 9 + t] = v;
    }
    __syncthreads();
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static bool dw_shape_ok(int C, int k) { return k == 3 && C >= 8 && (C % 8) == 0 && C <= 2048 && (DW_T % (C / 8)) == 0; }

extern "C" int pf_depthwise_supported(int C, int k, int stride) { return (dw_shape_ok(C, k) && (stride == 1 || stride == 2)) ? 1 : 0; }

static int dw_groups(int64_t total, int C) {
  const int spb = DW_T / (C / 8);
  int64_t g = (total + spb - 1) / spb;
  if (g > 1024) g = 1024;                                   // ~4 workgroups per CU, grid-stride beyond
  if (g < 1) g = 1;
  return (int)g;
}

// rows of the statistics array pf_depthwise_fwd writes ([G][4][C]) / slabs of pf_depthwise_wrw ([G][C][9] floats)
extern "C" int pf_depthwise_groups(int B, int Ho, int Wo, int C) {
  if (C < 8 || (C % 8)) return 0;
  const int64_t total = (int64_t)B * Ho * ((Wo + DW_PW - 1) / DW_PW);
  return dw_groups(total, C);
}

// sizes of the tensors the kernels read through buffer descriptors (31-bit byte offsets: 0x80000000 is the "outside" offset)
static bool dw_set_bytes(DwArgs& a, int dtype, int64_t x_elems, int64_t dy_elems) {
  const int64_t sz = (dtype == PF_F32) ? 4 : 2;
  if (x_elems * sz >= ((int64_t)1 << 31) || dy_elems * sz >= ((int64_t)1 << 31)) return false;
  a.x_bytes = (uint32_t)(x_elems * sz);
  a.dy_bytes = (uint32_t)(dy_elems * sz);
  return true;
}

static int dw_fill(DwArgs& a, int B, int H, int W, int C, int Ho, int Wo, int stride, int ph, int pw) {
  if (B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || ph < 0 || pw < 0 || ph > 2 || pw > 2) return (int)hipErrorInvalidValue;
  a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.ph = ph; a.pw = pw;
  a.flip = 0;
  a.strips_w = (Wo + DW_PW - 1) / DW_PW;
  a.total = (int64_t)B * Ho * a.strips_w;
  return 0;
}

extern "C" int pf_depthwise_fwd(const void* X, const void* W, void* Y, int dtype, float* partial, int B, int H, int Wd, int C,
                                int k, int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream) {
  if (!pf_depthwise_supported(C, k, stride)) return (int)hipErrorInvalidValue;
  if (!pf_aligned16(X) || !pf_aligned16(Y) || X == nullptr || W == nullptr || Y == nullptr) return (int)hipErrorInvalidValue;
  DwArgs a;
  const int r = dw_fill(a, B, H, Wd, C, Ho, Wo, stride, pad_h, pad_w);
  if (r) return r;
  a.x = X; a.w = W; a.y = Y; a.dy = nullptr; a.partial = partial;
  if (!dw_set_bytes(a, dtype, (int64_t)B * H * Wd * C, 0)) return (int)hipErrorInvalidValue;
  const int grid = dw_groups(a.total, C);
  hipStream_t st = (hipStream_t)stream;
#define PF_DW(TT)                                                                     \
  do {                                                                                \
    if (stride == 1) k_dw_fwd<TT, 1><<<grid, DW_T, 0, st>>>(a);                       \
    else k_dw_fwd<TT, 2><<<grid, DW_T, 0, st>>>(a);                                   \
  } while (0)
  if (dtype == PF_F32) PF_DW(float);
  else if (dtype == PF_BF16) PF_DW(bf16_t);
  else return (int)hipErrorInvalidValue;
#undef PF_DW
  PF_LAUNCH_CHECK();
  return 0;
}

// dX [B][H][Wd][C] from dY [B][Ho][Wo][C]; W is the forward kernel [C][3][3] (not flipped), pad_h / pad_w the forward front pads
extern "C" int pf_depthwise_bwd_data(const void* dY, const void* W, void* dX, int dtype, int B, int H, int Wd, int C, int k,
                                     int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream) {
  if (!pf_depthwise_supported(C, k, stride)) return (int)hipErrorInvalidValue;
  if (!pf_aligned16(dY) || !pf_aligned16(dX) || dY == nullptr || W == nullptr || dX == nullptr) return (int)hipErrorInvalidValue;
  DwArgs a;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) {
    // the same strip kernel on the flipped taps: dx[hi][wi] = sum_{r',s'} dy[hi + r' - (2-ph)][wi + s' - (2-pw)] * w[2-r'][2-s']
    const int r1 = dw_fill(a, B, Ho, Wo, C, H, Wd, 1, 2 - pad_h, 2 - pad_w);
    if (r1) return r1;
    a.x = dY; a.w = W; a.y = dX; a.dy = nullptr; a.partial = nullptr; a.flip = 1;
    if (!dw_set_bytes(a, dtype, (int64_t)B * Ho * Wo * C, 0)) return (int)hipErrorInvalidValue;
    const int grid1 = dw_groups(a.total, C);
    if (dtype == PF_F32) k_dw_fwd<float, 1><<<grid1, DW_T, 0, st>>>(a);
    else if (dtype == PF_BF16) k_dw_fwd<bf16_t, 1><<<grid1, DW_T, 0, st>>>(a);
    else return (int)hipErrorInvalidValue;
    PF_LAUNCH_CHECK();
    return 0;
  }
  const int r = dw_fill(a, B, H, Wd, C, Ho, Wo, stride, pad_h, pad_w);
  if (r) return r;
  a.x = dY; a.w = W; a.y = dX; a.dy = nullptr; a.partial = nullptr;
  if (!dw_set_bytes(a, dtype, (int64_t)B * Ho * Wo * C, 0)) return (int)hipErrorInvalidValue;
  const int64_t npix = (int64_t)B * H * Wd;
  const int grid = dw_groups(npix, C);
  if (dtype == PF_F32) k_dw_bwd_data<float><<<grid, DW_T, 0, st>>>(a);
  else if (dtype == PF_BF16) k_dw_bwd_data<bf16_t><<<grid, DW_T, 0, st>>>(a);
  else return (int)hipErrorInvalidValue;
  PF_LAUNCH_CHECK();
  return 0;
}

int pf_wrw_reduce(float* workspace, int S, int64_t n, void* dW, int dw_dtype, hipStream_t st);   // pf_conv.hip

// dW [C][3][3] (dw_dtype) from dY and X; slabs: (pf_depthwise_groups(B, Ho, Wo, C) + 32) * C * 9 floats of workspace
extern "C" int pf_depthwise_wrw(const void* dY, const void* X, void* dW, int dtype, int dw_dtype, float* slabs, int B, int H,
                                int Wd, int C, int k, int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream) {
  if (!pf_depthwise_supported(C, k, stride)) return (int)hipErrorInvalidValue;
  if (!pf_aligned16(dY) || !pf_aligned16(X) || dY == nullptr || X == nullptr || dW == nullptr || slabs == nullptr)
    return (int)hipErrorInvalidValue;
  DwArgs a;
  const int r = dw_fill(a, B, H, Wd, C, Ho, Wo, stride, pad_h, pad_w);
  if (r) return r;
  a.x = X; a.w = nullptr; a.y = nullptr; a.dy = dY; a.partial = slabs;
  if (!dw_set_bytes(a, dtype, (int64_t)B * H * Wd * C, (int64_t)B * Ho * Wo * C)) return (int)hipErrorInvalidValue;
  const int grid = dw_groups(a.total, C);
  hipStream_t st = (hipStream_t)stream;
#define PF_DWW(TT)                                                                    \
  do {                                                                                \
    if (stride == 1) k_dw_wrw<TT, 1><<<grid, DW_T, 0, st>>>(a);                       \
    else k_dw_wrw<TT, 2><<<grid, DW_T, 0, st>>>(a);                                   \
  } while (0)
  if (dtype == PF_F32) PF_DWW(float);
  else if (dtype == PF_BF16) PF_DWW(bf16_t);
  else return (int)hipErrorInvalidValue;
#undef PF_DWW
  const int n = C * 9;
  // The slabs go through the staged reduction of the other backward-filter kernels (pf_wrw_reduce: 64 outputs x 4 interleaved slab
  // lanes per workgroup, split ranges on gridDim.y, a second launch folds the ranges; fixed order).  Round 4's own reduction ran ONE
  // thread per output over all G <= 1024 slabs (271 us per launch, 14.6 % of the C3 step); measured in round 5's first GPU call
  // (profiles/r05_first_call_ab.txt): C3 13 602 -> 16 140 images/s with this one, the old kernel is gone.
  return pf_wrw_reduce(slabs, grid, (int64_t)n, dW, dw_dtype, st);
}
