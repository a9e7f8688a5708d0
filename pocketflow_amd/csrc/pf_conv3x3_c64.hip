// K12, the one geometry the per-tap implicit GEMM (pf_igemm.hip) handles worst: 3x3 / stride 1 / pad 1, 64 -> 64 channels on 56 x 56
// feature maps -- conv2 of ResNet-50's first stage (utils/external/resnet_model.py:92-103, 257-314): forward of the student (with the
// statistics epilogue) and of the teacher (with the consumer's inference-mode BN in the epilogue), backward-data (flipped kernel,
// BN-backward sums in the epilogue).  802 816 output pixels per launch at batch 256, 11 launches per step.
//
// Why its own kernel.  With 64 input channels a tap is ONE 64-channel k-step: the per-tap kernel re-stages the [128 pixels][128 B]
// input tile for each of the nine taps (9 x 16 KiB of LDS-DMA per 128 x 64 outputs: the fill, not the matrix pipe, sets its
// 130-150 us against a 24 us MFMA / 33 us HBM floor; profiles/r05_igemm_layers.txt).  Here
//   * a tile is TWO image rows (112 output pixels = 7 fragments of 16) x all 64 output channels; its input WINDOW -- 4 rows x 58
//     columns (the zero padding columns included) x 64 channels -- is staged ONCE (33 KiB of LDS-DMA instead of 9 x 14 KiB) and every
//     tap reads it at a shifted address: (r * 58 + s) * 144 bytes, an immediate offset of the fragment reads;
//   * window pixels are 144 bytes apart (128 + 16 of padding): sixteen consecutive pixels then cover all 64 LDS banks exactly once in
//     a 16-byte fragment read -- no XOR swizzle, which would depend on the tap;
//   * the whole kernel -- 64 x 576 bf16 = 72 KiB -- lives in REGISTERS for the life of the persistent workgroup: wavefront (wm, wn)
//     owns output channels wn * 32 .. + 32 (2 blocks x 18 k-blocks x 4 registers = 144) and pixel fragments {0..3} (wm = 0) or
//     {4..6} (wm = 1).  Per k-block a wavefront reads 4 (3) input fragments for 8 (6) MFMAs: 14 KiB of LDS reads per 28 MFMAs of
//     the workgroup, against 32 KiB per 32 MFMAs in the per-tap kernel's 64 x 32 wavefront tiles;
//   * two windows in LDS (2 x 33 KiB): the next tile's window travels while this one is multiplied; two workgroups per CU.
// Padding rows above / below the image and the two padding columns are LDS-DMA lanes with an out-of-range offset (zeros).
// The epilogue (C tile through LDS, row pass with statistics / BN-backward sums / output affine, fixed-order reduction of the
// per-workgroup statistics) is the one of pf_igemm.hip.
#include "pf_igemm.h"

#define H3_W 56                                  // image height = width
#define H3_C 64                                  // input channels = output channels
#define H3_TP 112                                // output pixels per tile: two image rows
#define H3_WCOLS 58
#define H3_PITCH 144                             // bytes per window pixel
#define H3_WPIX (4 * H3_WCOLS)                   // 232 window pixels
#define H3_NDMA 33                               // 1 KiB LDS-DMA instructions per window (232 x 144 B = 33 408 B)
#define H3_WBYTES (H3_NDMA * 1024)
#define H3_THREADS 256
#define H3_CS_LD (H3_C + 8)
#define H3_TILES_PER_IMG (H3_W / 2)

// fragment read with an immediate byte offset (tap shift + k-block half), WITHOUT its wait: see pf_conv_common.h for why LDS accesses
// beside a pending LDS-DMA come from inline asm
template <int OFF>
__device__ __forceinline__ void h3_read(uint32_t p, u32x4_t& v) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void h3_wait4(u32x4_t (&v)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
}
__device__ __forceinline__ void h3_wait3(u32x4_t (&v)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]) : : "memory");
}

// one k-block (32 channels of one tap) of a wavefront's tile: NJ input fragments, 2 x NJ MFMAs
template <int NJ, int KB>
__device__ __forceinline__ void h3_kblock(const uint32_t (&xb)[4], const bf16x8 (&w0)[18], const bf16x8 (&w1)[18], f32x4 (&acc)[2][4]) {
  constexpr int TAP = KB / 2, HALF = KB & 1;
  constexpr int OFF = ((TAP / 3) * H3_WCOLS + (TAP % 3)) * H3_PITCH + HALF * 64;
  u32x4_t x[4];
  h3_read<OFF>(xb[0], x[0]);
  h3_read<OFF>(xb[1], x[1]);
  h3_read<OFF>(xb[2], x[2]);
  if (NJ == 4) { h3_read<OFF>(xb[3], x[3]); h3_wait4(x); } else h3_wait3(x);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(&x[j]);
    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[KB], xf, acc[0][j], 0, 0, 0);
    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[KB], xf, acc[1][j], 0, 0, 0);
  }
}

template <int NJ>
__device__ __forceinline__ void h3_tile(const uint32_t (&xb)[4], const bf16x8 (&w0)[18], const bf16x8 (&w1)[18], f32x4 (&acc)[2][4]) {
  h3_kblock<NJ, 0>(xb, w0, w1, acc);   h3_kblock<NJ, 1>(xb, w0, w1, acc);   h3_kblock<NJ, 2>(xb, w0, w1, acc);
  h3_kblock<NJ, 3>(xb, w0, w1, acc);   h3_kblock<NJ, 4>(xb, w0, w1, acc);   h3_kblock<NJ, 5>(xb, w0, w1, acc);
  h3_kblock<NJ, 6>(xb, w0, w1, acc);   h3_kblock<NJ, 7>(xb, w0, w1, acc);   h3_kblock<NJ, 8>(xb, w0, w1, acc);
  h3_kblock<NJ, 9>(xb, w0, w1, acc);   h3_kblock<NJ, 10>(xb, w0, w1, acc);  h3_kblock<NJ, 11>(xb, w0, w1, acc);
  h3_kblock<NJ, 12>(xb, w0, w1, acc);  h3_kblock<NJ, 13>(xb, w0, w1, acc);  h3_kblock<NJ, 14>(xb, w0, w1, acc);
  h3_kblock<NJ, 15>(xb, w0, w1, acc);  h3_kblock<NJ, 16>(xb, w0, w1, acc);  h3_kblock<NJ, 17>(xb, w0, w1, acc);
}

// MODE: IG_PLAIN | IG_BWD (BN-backward sums); STATS: the statistics epilogue (IG_BWD: always); AFF: output affine of the row pass (see
// pf_igemm.hip).  STATS is a template parameter because its 24 accumulator registers do not fit beside the 144 of the kernel slice
// unless they are known dead; the per-channel minimum / maximum are kept as PACKED bf16 pairs (the values are bf16 numbers).
template <int MODE, bool STATS, bool AFF>
__global__ __launch_bounds__(H3_THREADS, 2) void k_conv3x3_c64(const IgArgs a) {
  constexpr bool BWD = (MODE == IG_BWD);
  static_assert(!BWD || STATS, "backward-data launches exist for their statistics");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // two windows; aliased: C tile, statistics scratch
  float* bpl = reinterpret_cast<float*>(smem + 2 * H3_WBYTES);              // BWD: scale | shift | mean | invstd [4][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, q = lane >> 4;
  const int n_tiles = (a.M / (H3_W * H3_W)) * H3_TILES_PER_IMG;

  // ---- the kernel slice of this wavefront, once: W[n][tap][c], k = tap * 64 + c; k-block kb = 32 consecutive k ------------------
  bf16x8 w0[18], w1[18];
  {
    const bf16_t* wp0 = a.W + (int64_t)(wn * 32 + l15) * (9 * H3_C) + q * 8;
    const bf16_t* wp1 = wp0 + 16 * (9 * H3_C);
#pragma unroll
    for (int kb = 0; kb < 18; ++kb) {
      w0[kb] = *reinterpret_cast<const bf16x8*>(wp0 + kb * 32);
      w1[kb] = *reinterpret_cast<const bf16x8*>(wp1 + kb * 32);
    }
  }
  if (BWD) {
    for (int i = tid; i < 4 * H3_C; i += H3_THREADS) {
      const int qq = i / H3_C, c = i - qq * H3_C;
      bpl[i] = (qq < 2) ? a.bss[qq * H3_C + c] : a.bmi[(qq - 2) * H3_C + c];
    }
  }

  // ---- LDS-DMA slots of this wavefront: instruction I = wave + 4 * d covers LDS bytes [I * 1024, + 1024) of a window; lane L its
  // bytes [L * 16, + 16) = 16-byte group ck of window pixel wp.  Tile-invariant: the source offset relative to the tile's first pixel
  // and which padding class the pixel belongs to (0: inside, 1: row above the tile, 2: row below it, 3: never loaded) ----------------
  const pf_rsrc_t rsX = PF_MAKE_RSRC(a.X, a.x_bytes);
  constexpr uint32_t OOB = 0x80000000u;
  // (recomputed per tile from the lane index -- a dozen integer operations per instruction -- instead of held in 10 registers: with the
  // kernel slice in 144 registers the tile loop has none to spare)
  auto stage = [&](int t, int buf) {
    const int img = t / H3_TILES_PER_IMG, h0 = (t - img * H3_TILES_PER_IMG) * 2;
    const int base = (img * (H3_W * H3_W) + h0 * H3_W) * (H3_C * 2);
    const bool top_ok = h0 > 0, bot_ok = h0 + 2 < H3_W;
    unsigned char* dst = smem + buf * H3_WBYTES;
#pragma unroll
    for (int d = 0; d < 9; ++d) {
      const int I = wave + 4 * d;                                            // wave-uniform
      if (I < H3_NDMA) {
        const int byte = I * 1024 + lane * 16;
        const int wp = byte / H3_PITCH, ck = (byte - wp * H3_PITCH) >> 4;
        const int wr = wp / H3_WCOLS, wc = wp - wr * H3_WCOLS;
        const bool ok = wp < H3_WPIX && ck < 8 && wc > 0 && wc < H3_WCOLS - 1 && (wr > 0 || top_ok) && (wr < 3 || bot_ok);
        const uint32_t voff = ok ? (uint32_t)(base + ((wr - 1) * H3_W + (wc - 1)) * (H3_C * 2) + ck * 16) : OOB;
        PF_BUFFER_LOAD_LDS16(rsX, dst + I * 1024, voff, 0);
      }
    }
  };

  // ---- fragment bases: pixel p = f * 16 + l15 of the tile reads window pixel (p / 56) * 58 + p % 56 (+ the tap's shift) ---------
  uint32_t xb[4];                                                            // ... in window 0; toggled to the other window per tile
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int p = (wm * 4 + j) * 16 + l15;
    p = (p < H3_TP) ? p : (H3_TP - 1);                                       // (wm = 1, j = 3: not a fragment of the tile; never read)
    const int wp0 = (p / H3_W) * H3_WCOLS + (p % H3_W);
    xb[j] = lds_addr(smem) + (uint32_t)(wp0 * H3_PITCH + q * 16);
  }

  float st_s[8], st_q[8];
  uint32_t st_mn[4], st_mx[4];                                               // packed bf16 pairs (+inf / -inf)
#pragma unroll
  for (int j = 0; j < 8; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
#pragma unroll
  for (int j = 0; j < 4; ++j) { st_mn[j] = 0x7F807F80u; st_mx[j] = 0xFF80FF80u; }
  constexpr int VPR = H3_C / 8, RPP = H3_THREADS / VPR, NP = (H3_TP + RPP - 1) / RPP;   // 8 vectors per row, 32 rows per pass, 4 passes
  const int wvec = tid % VPR, wrw = tid / VPR;
  bf16_t* Cs;

  int buf = 0;
  if ((int)blockIdx.x < n_tiles) stage(blockIdx.x, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < n_tiles) stage(tn, buf ^ 1);                                    // travels under this tile's matrix work
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (wm == 0) h3_tile<4>(xb, w0, w1, acc); else h3_tile<3>(xb, w0, w1, acc);
    __builtin_amdgcn_s_barrier();                                            // every wavefront is done reading this window
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: C tile through the window that was just consumed (its LDS is free until the NEXT iteration stages into it) ----
    Cs = reinterpret_cast<bf16_t*>(smem + buf * H3_WBYTES);
    const int nj = (wm == 0) ? 4 : 3;
    const int m0 = t * H3_TP;
    if (!BWD && a.R != nullptr) {                                            // residual on the fp32 accumulators: ONE rounding to bf16
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nj) {
          const int m = m0 + (wm * 4 + j) * 16 + l15;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const uint2 r = *reinterpret_cast<const uint2*>(a.R + (int64_t)m * H3_C + wn * 32 + i * 16 + q * 4);
            acc[i][j][0] += __uint_as_float(r.x << 16); acc[i][j][1] += __uint_as_float(r.x & 0xFFFF0000u);
            acc[i][j][2] += __uint_as_float(r.y << 16); acc[i][j][3] += __uint_as_float(r.y & 0xFFFF0000u);
          }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < nj) {
          const uint2 v = make_uint2(pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3]));
          lds_write_b64(lds_addr(Cs + ((wm * 4 + j) * 16 + l15) * H3_CS_LD + wn * 32 + i * 16 + q * 4), v);
        }
      }
    uint4 rres[NP];
    if (BWD) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int rl = wrw + p * RPP;
        rres[p] = make_uint4(0, 0, 0, 0);
        if (rl < H3_TP) rres[p] = *reinterpret_cast<const uint4*>(a.bx + (int64_t)(m0 + rl) * H3_C + wvec * 8);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int rl = wrw + p * RPP;
      if (rl < H3_TP) {
        uint4 c = lds_read_b128(lds_addr(Cs + rl * H3_CS_LD + wvec * 8));
        const int n = wvec * 8;
        if (BWD) {
          // this thread's 8 channels of scale | shift | mean | invstd: eight 16-byte LDS reads per row (element by element they were 32
          // reads per row; held in registers across the tile they do not fit beside the kernel slice)
          float bpr[32];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const float4 v0 = *reinterpret_cast<const float4*>(bpl + qq * H3_C + n), v1 = *reinterpret_cast<const float4*>(bpl + qq * H3_C + n + 4);
            bpr[qq * 8 + 0] = v0.x; bpr[qq * 8 + 1] = v0.y; bpr[qq * 8 + 2] = v0.z; bpr[qq * 8 + 3] = v0.w;
            bpr[qq * 8 + 4] = v1.x; bpr[qq * 8 + 5] = v1.y; bpr[qq * 8 + 6] = v1.z; bpr[qq * 8 + 7] = v1.w;
          }
          float f[8], xv[8];
          unpack8(c, f);
          unpack8(rres[p], xv);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float u = fmaf(bpr[j], xv[j], bpr[8 + j]);
            const float dy = (u > a.b_lo && u < a.b_hi) ? f[j] : 0.f;
            st_s[j] += dy;
            st_q[j] = fmaf(dy, (xv[j] - bpr[16 + j]) * bpr[24 + j], st_q[j]);
          }
        } else if constexpr (STATS) {
          float f[8];
          unpack8(c, f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            st_s[j] += f[j];
            st_q[j] = fmaf(f[j], f[j], st_q[j]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float mn0 = pf_acc_min(__uint_as_float(st_mn[j] << 16), f[2 * j]), mn1 = pf_acc_min(__uint_as_float(st_mn[j] & 0xFFFF0000u), f[2 * j + 1]);
            const float mx0 = pf_acc_max(__uint_as_float(st_mx[j] << 16), f[2 * j]), mx1 = pf_acc_max(__uint_as_float(st_mx[j] & 0xFFFF0000u), f[2 * j + 1]);
            st_mn[j] = (__float_as_uint(mn0) >> 16) | (__float_as_uint(mn1) & 0xFFFF0000u);     // exact: every operand is a bf16 number
            st_mx[j] = (__float_as_uint(mx0) >> 16) | (__float_as_uint(mx1) & 0xFFFF0000u);
          }
        }
        if constexpr (AFF) c = out_affine8(c, a.oss, H3_C, n, a.oact);
        *reinterpret_cast<uint4*>(a.Y + (int64_t)(m0 + rl) * H3_C + n) = c;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // the next window has landed (and this tile's stores are out)
    __builtin_amdgcn_s_barrier();                                            // ... for every wavefront; the C tile is free again
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) xb[j] = buf ? xb[j] - (uint32_t)H3_WBYTES : xb[j] + (uint32_t)H3_WBYTES;
    buf ^= 1;
  }

  // ---- per-workgroup statistics -> partial[g][stat][64] (fixed order: deterministic), as pf_igemm.hip ----------------------------
  if constexpr (STATS) {
    float* red = reinterpret_cast<float*>(smem);
    const int nstat = BWD ? 2 : 4;
    __syncthreads();
    for (int stat = 0; stat < nstat; ++stat) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v;
        if (stat == 0) v = st_s[j];
        else if (stat == 1) v = st_q[j];
        else {
          const uint32_t pk = (stat == 2) ? st_mn[j >> 1] : st_mx[j >> 1];
          v = __uint_as_float((j & 1) ? (pk & 0xFFFF0000u) : (pk << 16));
        }
        red[wrw * H3_C + wvec * 8 + j] = v;
      }
      __syncthreads();
      for (int c = tid; c < H3_C; c += H3_THREADS) {
        float r = red[c];
        for (int rr = 1; rr < RPP; ++rr) {
          const float w = red[rr * H3_C + c];
          r = (stat < 2) ? (r + w) : (stat == 2 ? fminf(r, w) : fmaxf(r, w));
        }
        a.partial[((int64_t)blockIdx.x * nstat + stat) * H3_C + c] = r;
      }
      __syncthreads();
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
static bool h3_enabled() { return pf_tuning().conv3x3_c64 != 0; }      // PF_CONV3X3_C64=0: the per-tap kernel (A/B runs, tests)

// the geometry this kernel is written for (what pf_conv2d_stats_groups_geom can see of a launch) ...
bool pf_conv3x3_c64_geom(int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w, int Ho, int Wo) {
  return h3_enabled() && pf_tuning().igemm_tile_bm == 0 && th == 3 && tw == 3 && stride == 1 && pad_h == 1 && pad_w == 1 &&
         C == H3_C && N == H3_C && H == H3_W && Wd == H3_W && Ho == H3_W && Wo == H3_W;
}

// ... and whether it takes THIS launch (a: filled by pf_igemm.hip's entry points): no prologue, no sub-filter walk
bool pf_conv3x3_c64_takes(const IgArgs& a) {
  return pf_conv3x3_c64_geom(a.H, a.Wd, a.C, a.N, a.th, a.tw, a.stride, a.pad_h, a.pad_w, a.Ho, a.Wo) && a.ss == nullptr &&
         a.o_sub == 0 && a.w_taps_full == 9 && a.w_rs == 1 && a.w_ss == 1 && a.w_r0 == 0 && a.w_s0 == 0 &&
         a.M % (H3_W * H3_W) == 0 && !(a.oss != nullptr && (a.R != nullptr || a.partial != nullptr));
}

static int h3_grid(int M) {
  const int n_tiles = (M / (H3_W * H3_W)) * H3_TILES_PER_IMG;
  return n_tiles < 512 ? n_tiles : 512;                     // two persistent workgroups per CU
}

int pf_conv3x3_c64_stats_groups(int M) { return h3_grid(M); }

template <int MODE, bool STATS, bool AFF>
static int h3_launch_t(const IgArgs& a, hipStream_t st) {
  const size_t lds = 2 * (size_t)H3_WBYTES + (MODE == IG_BWD ? 4 * H3_C * 4 : 0);
  if (int e = pf_require_lds(reinterpret_cast<const void*>(&k_conv3x3_c64<MODE, STATS, AFF>), lds)) return e;
  k_conv3x3_c64<MODE, STATS, AFF><<<h3_grid(a.M), H3_THREADS, lds, st>>>(a);
  PF_LAUNCH_CHECK();
  return 0;
}

int pf_conv3x3_c64_launch(const IgArgs& a, hipStream_t st) {
  if (a.bx != nullptr) return h3_launch_t<IG_BWD, true, false>(a, st);
  if (a.oss != nullptr) return h3_launch_t<IG_PLAIN, false, true>(a, st);
  if (a.partial != nullptr) return h3_launch_t<IG_PLAIN, true, false>(a, st);
  return h3_launch_t<IG_PLAIN, false, false>(a, st);
}
