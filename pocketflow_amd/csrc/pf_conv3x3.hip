// K12: 3x3 / stride 1 / pad 1 convolutions (the middle convolution of every ResNet bottleneck block and both convolutions of
// the basic blocks: conv2d_fixed_padding, utils/external/resnet_model.py:92-103), forward and -- on the flipped / transposed
// kernel -- backward-data, bf16 NHWC activations, KRSC kernels, float32 accumulation.  Same operation and epilogues as
// k_igemm (pf_igemm.hip): Y = conv(X, W) [+ R], per-channel statistics of Y for the consumer BN, or the BN-backward sums of
// the producer BN (backward-data).
//
// Why a second kernel.  Round 3's ablations (tools/gpu/igemm_ablate.py, profiles/r03_igemm_ablation.txt) showed the implicit
// GEMM bound by the LDS FILL: with the LDS-DMA alone (no fragment reads, no MFMAs) the 3x3 launches take 75-85 % of their
// full time, and the fill rate saturates at 12-18 TB/s (L2 -> LDS) whatever the tile.  k_igemm re-fetches the input tile once
// per TAP: 9 x (input + kernel) tiles per 64 channels.  Here the input is staged ONCE per 64-channel chunk, with its halo, and
// the nine taps read it at nine different row offsets:
//
//   * Padded-flat pixel space.  Image i occupies rows hp = 0..H of (H+1) x (W+1) positions: hp = 0 is a zero row (the top
//     padding of image i = the bottom padding of image i-1), wp = 0 a zero column (left padding of a row = right padding of
//     the previous one); flat index f = (i * (H+1) + hp) * (W+1) + wp.  The output pixel (ho, wo) sits at f = (i, ho+1, wo+1),
//     and tap (r, s) reads f + (r-1) * (W+1) + (s-1): a PURE SHIFT of the flat index, padding included.  A tile of BM
//     consecutive flat positions therefore needs the input window [f0 - (W+1) - 1, f0 + BM + (W+1) + 1) -- staged once per
//     chunk by LDS-DMA (border positions read zeros through the buffer range check) -- and tap (r, s) is the row offset
//     r * (W+1) + s inside that window.  Border positions are computed and not stored: 3.5 % (56x56) ... 23 % (7x7) of the
//     MFMA work, bought for a 3.2x smaller fill (256 x 128 tile: 24 KB per step instead of 48 KB, and no per-tap gather).
//   * 256 x 128 tile, 8 wavefronts (64 x 64 accumulators each), one workgroup per CU; window double-buffered across chunks
//     (one 64-row pass is issued per step, so every step carries the same number of LDS-DMA instructions and the counted
//     vmcnt stays a constant), kernel tiles in a 3-stage ring; one barrier per (tap, chunk) step.
//   * weights = MFMA operand A, pixels = operand B (a lane's four accumulators are four consecutive output channels of one
//     pixel), XOR-swizzled 16-byte chunks as in k_igemm: the swizzle key of a window row is (row & 7), recomputed per tap.
#include "pf_conv_common.h"
#include <stdlib.h>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t pf_hrsrc_t;
#define PF_H_MAKE_RSRC(p, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(p), (short)0, (int)(bytes), 0x00020000)
#define PF_H_LOAD_LDS16(rs, lds, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, voff, soff, 0, 0)
#else
typedef int pf_hrsrc_t;                                 // host pass: the kernel body only has to parse (see pf_igemm.hip)
#define PF_H_MAKE_RSRC(p, bytes) 0
#define PF_H_LOAD_LDS16(rs, lds, voff, soff) ((void)(rs), (void)(lds), (void)(voff), (void)(soff))
#endif

struct H3Args {
  const bf16_t* X;      // [imgs][H][W][C]
  const bf16_t* W;      // [N][3][3][C]
  bf16_t* Y;            // [imgs][H][W][N]
  const bf16_t* R;      // residual [M][N] or null
  float* partial;       // statistics [G][4][N] ([G][2][N] with bx) or null
  const bf16_t* bx;     // BN-backward statistics mode: the BN's input x [M][N]
  const float* bss;     // its scale | shift [2][N]
  const float* bmi;     // its mean | invstd [2][N]
  float b_lo, b_hi;
  uint32_t x_bytes, w_bytes;
  int imgs, H, Wd, C, N;
  int F;                // flat positions: imgs * (H+1) * (Wd+1)
  int tiles_m, tiles_n, G;
};

template <int N> __device__ __forceinline__ void h3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#define H3_BM 256
#define H3_T 512
#define H3_WROWS 384                                    // window rows reserved per buffer (BM + 2 * (W+1) + 2 <= 384 <=> W <= 62)
#define H3_NPASS (H3_WROWS / 64)                        // 64-row LDS-DMA passes per window

template <int BN, bool BWD>
__global__ __launch_bounds__(H3_T) void k_conv3x3_halo(const H3Args a) {
  constexpr int BM = H3_BM, T = H3_T;
  constexpr int WM = 4, WN = 2;
  constexpr int WR = BM / WM, WC = BN / WN;            // 64 x (64 | 32) per wavefront
  constexpr int JM = WR / 16, NI = WC / 16;
  constexpr int BS = BN * 8 / T;                       // kernel-tile LDS-DMA instructions per lane and step (BN = 128: 2, 64: 1)
  constexpr int LPS = BS + 1;                          // ... plus one window pass: the constant the counted vmcnt needs
  constexpr int WIN_B = H3_WROWS * 128, WT_B = BN * 128, NSW = 3;
  constexpr int CS_LD = BN + 8;
  constexpr int VPR = BN / 8, RPP = T / VPR, NP = BM / RPP;
  static_assert(BS >= 1 && NP >= 1, "tile too small for the block");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [window 0][window 1][kernel ring: 3 tiles][sink 1 KiB][BWD vectors]; the C tile and the statistics scratch alias the windows
  unsigned char* const win0 = smem;
  unsigned char* const wring = smem + 2 * WIN_B;
  unsigned char* const sink = wring + NSW * WT_B;
  float* const bpl = reinterpret_cast<float*>(sink + 1024);                  // BWD: scale | shift | mean | invstd [4][BN]
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, q = lane >> 4;
  const int srow = tid >> 3;                                                // staging row inside a 64-row pass
  const int schunk = (lane & 7) ^ ((lane >> 3) & 7);

  const int xcd = blockIdx.x & 7, L = blockIdx.x >> 3;
  const int g = xcd + 8 * (L / a.tiles_n), tn = L % a.tiles_n;
  const int n0 = tn * BN;
  const int cch = a.C >> 6;
  const int nk = 9 * cch;
  const int Wp = a.Wd + 1, HWp = (a.H + 1) * Wp;
  const int64_t wrow = (int64_t)9 * a.C;

  if (BWD) {
    for (int i = tid; i < 4 * BN; i += T) {
      const int qq = i / BN, c = n0 + (i - qq * BN);
      float v = 0.f;
      if (c < a.N) v = (qq < 2) ? a.bss[qq * a.N + c] : a.bmi[(qq - 2) * a.N + c];
      bpl[i] = v;
    }
  }
  float st_s[8], st_q[8], st_mn[8], st_mx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; st_mn[j] = INFINITY; st_mx[j] = -INFINITY; }
  const int wvec = tid % VPR, wrw = tid / VPR;
  const bf16_t* __restrict__ side = BWD ? a.bx : nullptr;                 // the residual is added on the accumulators

  const pf_hrsrc_t rsX = PF_H_MAKE_RSRC(a.X, a.x_bytes);
  const pf_hrsrc_t rsW = PF_H_MAKE_RSRC(a.W, a.w_bytes);
  constexpr uint32_t OOB = 0x80000000u;
  uint32_t boff[BS];
#pragma unroll
  for (int i = 0; i < BS; ++i) {
    const int n = n0 + i * (T / 8) + srow;
    boff[i] = (n < a.N) ? (uint32_t)(((int64_t)n * wrow + schunk * 8) * 2) : OOB;
  }

  for (int tm = g; tm < a.tiles_m; tm += a.G) {
    const int f0 = tm * BM;
    const int fw0 = f0 - Wp - 1;                                            // flat index of window row 0
    const int win_rows = BM + 2 * Wp + 2;
    // source offsets of this lane's window rows (the same for every chunk: the chunk enters through the scalar offset)
    uint32_t wbase[H3_NPASS];
#pragma unroll
    for (int p = 0; p < H3_NPASS; ++p) {
      const int j = p * 64 + srow;
      const int f = fw0 + j;
      wbase[p] = OOB;
      if (j < win_rows && f >= 0 && f < a.F) {
        const int img = f / HWp, rem = f - img * HWp;
        const int hp = rem / Wp, wp = rem - hp * Wp;
        if (hp > 0 && wp > 0) wbase[p] = (uint32_t)((((img * a.H + hp - 1) * a.Wd + wp - 1) * a.C + schunk * 8) * 2);
      }
    }
    auto stage_win_pass = [&](int buf, int p, int cc) {                     // pass p of window chunk cc (p >= NPASS_LIVE: dummy)
      unsigned char* dst = (p < H3_NPASS) ? win0 + buf * WIN_B + (p * 64 + wave * 8) * 128 : sink;
      uint32_t v = OOB;
#pragma unroll
      for (int pp = 0; pp < H3_NPASS; ++pp) v = (pp == p) ? wbase[pp] : v;
      PF_H_LOAD_LDS16(rsX, dst, v, cc * 128);
    };
    auto stage_w = [&](int slot, int tap, int cc) {
      unsigned char* dst = wring + slot * WT_B;
#pragma unroll
      for (int i = 0; i < BS; ++i)
        PF_H_LOAD_LDS16(rsW, dst + (i * (T / 8) + wave * 8) * 128, boff[i], (tap * a.C + cc * 64) * 2);
    };

    f32x4 acc[NI][JM];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < JM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: the whole window of chunk 0, kernel tiles of steps 0 and 1 ------------------------------------------
#pragma unroll
    for (int p = 0; p < H3_NPASS; ++p) stage_win_pass(0, p, 0);
    stage_w(0, 0, 0);
    stage_w(1, 1, 0);                      // step order: chunk-major -- step s = (chunk s / 9, tap s % 9); nk >= 9
    h3_wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    int wslot = 0, islot = 2;
    int s_tap = 0, s_cc = 0;                                                // (tap, chunk) of the step being multiplied
    int i_tap = 2, i_cc = 0;                                                // ... of the kernel tile issued next (step s + 2)
    for (int s = 0; s < nk; ++s) {
      // issue: kernel tile of step s+2, and pass s_tap of the NEXT chunk's window (passes >= H3_NPASS / chunks past the end:
      // a dummy load into the sink, so that every step issues exactly LPS LDS-DMA instructions)
      if (s + 2 < nk) {
        stage_w(islot, i_tap, i_cc);
        islot = (islot + 1 == NSW) ? 0 : islot + 1;
        if (++i_tap == 9) { i_tap = 0; ++i_cc; }
      } else {
#pragma unroll
        for (int i = 0; i < BS; ++i) PF_H_LOAD_LDS16(rsW, sink, OOB, 0);
      }
      {
        const bool live = s_cc + 1 < cch && s_tap < H3_NPASS;
        const int p = live ? s_tap : H3_NPASS;
        stage_win_pass((s_cc + 1) & 1, p, s_cc + 1);
      }
      // multiply step s: kernel tile in ring slot wslot, pixels = window rows shifted by r * Wp + s
      const unsigned char* Ws = wring + wslot * WT_B;
      const unsigned char* Win = win0 + (s_cc & 1) * WIN_B;
      wslot = (wslot + 1 == NSW) ? 0 : wslot + 1;
      const int r = s_tap / 3, sx = s_tap - r * 3;
      const int shift = r * Wp + sx;
      const int prow = l15 + shift;                                          // window row of this lane's pixel in fragment block 0
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int coffw = (((kk * 4 + q) ^ (l15 & 7)) << 4);
        const int coffx = (((kk * 4 + q) ^ (prow & 7)) << 4);
        bf16x8 wf[NI], xf[JM];
#pragma unroll
        for (int i = 0; i < NI; ++i)
          wf[i] = *reinterpret_cast<const bf16x8*>(Ws + (wn * WC + i * 16 + l15) * 128 + coffw);
#pragma unroll
        for (int j = 0; j < JM; ++j)
          xf[j] = *reinterpret_cast<const bf16x8*>(Win + (wm * WR + j * 16 + prow) * 128 + coffx);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < JM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
      if (++s_tap == 9) { s_tap = 0; ++s_cc; }
      __builtin_amdgcn_sched_barrier(0);
      h3_wait_vm<LPS>();                                                    // everything but this step's own issues has landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    h3_wait_vm<0>();                                                        // the trailing dummy loads (the sink is never read)

    // ---- epilogue of one [BM][BN] tile; rows are flat positions: border positions are skipped ---------------------------
    if (!BWD && a.R != nullptr) {
      // residual on the fp32 accumulators (one rounding, in the staging below); this lane's rows are flat positions
#pragma unroll
      for (int j = 0; j < JM; ++j) {
        const int f = f0 + wm * WR + j * 16 + l15;
        int m = -1;
        if (f < a.F) {
          const int img = f / HWp, rem = f - img * HWp;
          const int hp = rem / Wp, wp = rem - hp * Wp;
          if (hp > 0 && wp > 0) m = (img * a.H + hp - 1) * a.Wd + wp - 1;
        }
        if (m >= 0) {
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const int n = n0 + wn * WC + i * 16 + q * 4;
            if (n < a.N) {
              const uint2 r = *reinterpret_cast<const uint2*>(a.R + (int64_t)m * a.N + n);
              acc[i][j][0] += __uint_as_float(r.x << 16); acc[i][j][1] += __uint_as_float(r.x & 0xFFFF0000u);
              acc[i][j][2] += __uint_as_float(r.y << 16); acc[i][j][3] += __uint_as_float(r.y & 0xFFFF0000u);
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < JM; ++j) {
        const uint2 v = make_uint2(pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3]));
        *reinterpret_cast<uint2*>(Cs + (wm * WR + j * 16 + l15) * CS_LD + wn * WC + i * 16 + q * 4) = v;
      }
    // output rows of this thread: flat position -> pixel index m (or -1)
    int mrow[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int f = f0 + wrw + p * RPP;
      mrow[p] = -1;
      if (f < a.F) {
        const int img = f / HWp, rem = f - img * HWp;
        const int hp = rem / Wp, wp = rem - hp * Wp;
        if (hp > 0 && wp > 0) mrow[p] = (img * a.H + hp - 1) * a.Wd + wp - 1;
      }
    }
    constexpr int PG = (NP > 4) ? 4 : NP;
    uint4 rres[PG];
    auto load_side = [&](int p0) {
#pragma unroll
      for (int p = 0; p < PG; ++p) {
        const int n = n0 + wvec * 8;
        rres[p] = make_uint4(0, 0, 0, 0);
        int m = -1;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) m = (pp == p0 + p) ? mrow[pp] : m;
        if (m >= 0 && n < a.N) rres[p] = *reinterpret_cast<const uint4*>(side + (int64_t)m * a.N + n);
      }
    };
    if (side != nullptr) load_side(0);
    __syncthreads();
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += PG) {
#pragma unroll
      for (int pp = 0; pp < PG; ++pp) {
        const int p = p0 + pp;
        const int rl = wrw + p * RPP;
        const int m = mrow[p], n = n0 + wvec * 8;
        if (m >= 0 && n < a.N) {
          uint4 c = *reinterpret_cast<const uint4*>(Cs + rl * CS_LD + wvec * 8);
          if (BWD) {
            float f[8], xv[8];
            unpack8(c, f);
            unpack8(rres[pp], xv);
            const float* bp = bpl + wvec * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float u = fmaf(bp[j], xv[j], bp[BN + j]);
              const float dy = (u > a.b_lo && u < a.b_hi) ? f[j] : 0.f;
              st_s[j] += dy;
              st_q[j] = fmaf(dy, (xv[j] - bp[2 * BN + j]) * bp[3 * BN + j], st_q[j]);
            }
          } else if (a.R != nullptr || a.partial != nullptr) {
            float f[8];
            unpack8(c, f);
            if (a.partial != nullptr) {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                st_s[j] += f[j];
                st_q[j] = fmaf(f[j], f[j], st_q[j]);
                st_mn[j] = pf_acc_min(st_mn[j], f[j]);
                st_mx[j] = pf_acc_max(st_mx[j], f[j]);
              }
            }
          }
          *reinterpret_cast<uint4*>(a.Y + (int64_t)m * a.N + n) = c;
        }
      }
      if (side != nullptr && p0 + PG < NP) load_side(p0 + PG);
    }
    __syncthreads();
  }

  // ---- per-workgroup statistics -> partial[g][stat][N] (fixed order: deterministic) ---------------------------
  if (a.partial != nullptr) {
    const int nstat = BWD ? 2 : 4;
    __syncthreads();
    for (int stat = 0; stat < nstat; ++stat) {
      const float* v = (stat == 0) ? st_s : (stat == 1 ? st_q : (stat == 2 ? st_mn : st_mx));
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wrw * BN + wvec * 8 + j] = v[j];
      __syncthreads();
      for (int c = tid; c < BN; c += T) {
        float r = red[c];
        for (int rr = 1; rr < RPP; ++rr) {
          const float w = red[rr * BN + c];
          r = (stat < 2) ? (r + w) : (stat == 2 ? fminf(r, w) : fmaxf(r, w));
        }
        if (n0 + c < a.N) a.partial[((int64_t)g * nstat + stat) * a.N + n0 + c] = r;
      }
      __syncthreads();
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static bool h3_enabled() {
  // =1: this kernel; default: the per-tap implicit GEMM of pf_igemm.hip.  Measured in round 3 (profiles/r03_igemm_layers.txt,
  // B = 256, us, halo vs 128 x 128 igemm with two workgroups per CU): 56x56 C=64 135 vs 136, 28x28 C=128 92 vs 85, 14x14 C=256
  // 82 vs 74, 7x7 C=512 72 vs 65.  A 3.2x smaller LDS fill did NOT make it faster: with one 8-wavefront workgroup per CU
  // every wavefront runs issue -> fragment reads -> MFMAs -> wait -> barrier in lockstep and the matrix pipe idles through
  // the first and the last two; two independent 4-wavefront workgroups per CU overlap those phases by drifting apart.
  // Kept opt-in (tests run it): the structure it needs next is a ping-pong schedule of the two wavefronts of a SIMD.
  return pf_tuning().conv3x3_halo != 0;                   // PF_CONV3X3_HALO
}

// does the halo kernel take this convolution?  (3x3, stride 1, pad 1, same size; 64-channel chunks; window fits the LDS)
int pf_conv3x3_halo_ok(int imgs, int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w, int Ho, int Wo) {
  if (!h3_enabled()) return 0;
  if (th != 3 || tw != 3 || stride != 1 || pad_h != 1 || pad_w != 1 || Ho != H || Wo != Wd) return 0;
  if ((C % 64) || (N % 64) || H3_BM + 2 * (Wd + 1) + 2 > H3_WROWS) return 0;
  if ((int64_t)imgs * (H + 1) * (Wd + 1) >= ((int64_t)1 << 30)) return 0;
  return 1;
}

static int h3_bn(int N) { return (N % 128 == 0) ? 128 : 64; }

static int h3_grid(int tiles_m, int tiles_n, int* G_out) {
  int G = 256 / tiles_n;                                   // one workgroup per CU
  G = (G / 8) * 8;
  if (G < 8) G = 8;
  const int need = ((tiles_m + 7) / 8) * 8;
  if (G > need) G = need;
  *G_out = G;
  return G * tiles_n;
}

int pf_conv3x3_halo_groups(int imgs, int H, int Wd, int N) {
  const int F = imgs * (H + 1) * (Wd + 1);
  const int bn = h3_bn(N);
  int G;
  h3_grid((F + H3_BM - 1) / H3_BM, (N + bn - 1) / bn, &G);
  return G;
}

template <int BN, bool BWD>
static int h3_launch_t(H3Args& a, hipStream_t st) {
  a.tiles_m = (a.F + H3_BM - 1) / H3_BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const int grid = h3_grid(a.tiles_m, a.tiles_n, &a.G);
  const size_t lds = 2 * (size_t)H3_WROWS * 128 + 3 * (size_t)BN * 128 + 1024 + (BWD ? 4 * BN * 4 : 0);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_halo<BN, BWD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  k_conv3x3_halo<BN, BWD><<<grid, H3_T, lds, st>>>(a);
  PF_LAUNCH_CHECK();
  return 0;
}

int pf_conv3x3_halo_launch(const void* X, const void* W, void* Y, const void* R, float* partial, const void* bn_x,
                           const float* bss, const float* bmi, float b_lo, float b_hi, int imgs, int H, int Wd, int C, int N,
                           hipStream_t st) {
  H3Args a;
  a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.Y = (bf16_t*)Y; a.R = (const bf16_t*)R; a.partial = partial;
  a.bx = (const bf16_t*)bn_x; a.bss = bss; a.bmi = bmi; a.b_lo = b_lo; a.b_hi = b_hi;
  a.x_bytes = (uint32_t)((int64_t)imgs * H * Wd * C * 2);
  a.w_bytes = (uint32_t)((int64_t)N * 9 * C * 2);
  a.imgs = imgs; a.H = H; a.Wd = Wd; a.C = C; a.N = N;
  a.F = imgs * (H + 1) * (Wd + 1);
  const bool bwd = bn_x != nullptr;
  if (h3_bn(N) == 128) return bwd ? h3_launch_t<128, true>(a, st) : h3_launch_t<128, false>(a, st);
  return bwd ? h3_launch_t<64, true>(a, st) : h3_launch_t<64, false>(a, st);
}
