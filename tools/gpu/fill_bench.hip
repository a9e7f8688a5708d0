// tools/gpu/fill_bench.hip -- how fast can a CU be FILLED?  (tool only; not part of libpocketflow_hip.so)
//
// DESIGN.md section 9 (resource model of the 3x3 main loop): the implicit-GEMM kernels are bound by the rate at which bytes
// arrive in LDS through `buffer_load ... lds` (LDS-DMA): 14-18 TB/s chip-wide in the fill-only ablation, 37-45 % of the
// 64 B/clk/CU an XCD's L2 can deliver.  This program measures that rate ALONE -- no MFMA, optionally the fragment reads of the
// product beside it -- as a function of what a kernel author can choose:
//   working set   1 MiB ... 1 GiB    L2-resident (4 MiB per XCD), MALL-resident (256 MiB), HBM
//   pattern       0 = every wave-instruction fetches one contiguous KiB
//                 1 = the product's gather: 8 lanes per 128-byte row, 8 rows per wave-instruction, rows `row_stride` bytes apart
//                 2 = pattern 1 with the product's XOR swizzle of the 16-byte groups
//   bytes / lane  16 (`buffer_load_dwordx4 ... lds`) or 4 (`buffer_load_dword ... lds`)
//   cache policy  default or nt (aux = 2)
//   depth         stages in flight (2 | 3), pieces per stage, wavefronts per workgroup (4 | 8), workgroups per CU (1 | 2)
//   barrier       one s_barrier per stage (as the product) or none
//   reads         16 ds_read_b128 per wavefront and stage beside the fill (the fragment reads of a 64 x 64 wave tile) or none
//   transport     LDS-DMA, or 16-byte global loads into registers + ds_write_b128 one stage later (k_fill_reg)
// Output: one line per configuration: GB/s chip-wide and bytes / clock / CU.
//
//   build + run:  tools/gpu/fill_bench.sh        (hipcc --offload-arch=gfx950; needs the GPU)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t fb_rsrc_t;
#define FB_MAKE_RSRC(p, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(p), (short)0, (int)(bytes), 0x00020000)
#define FB_LOAD_LDS(rs, lds, SZ, voff, soff, AUX) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), SZ, voff, soff, 0, AUX)
#else
typedef int fb_rsrc_t;
#define FB_MAKE_RSRC(p, bytes) 0
#define FB_LOAD_LDS(rs, lds, SZ, voff, soff, AUX) ((void)(rs), (void)(lds), (void)(voff), (void)(soff))
#endif

struct FbArgs {
  const unsigned char* src;
  uint32_t bytes;        // working set, a power of two, <= 2 GiB (buffer descriptor range)
  uint32_t row_stride;   // patterns 1 / 2: bytes between gathered rows (a power of two >= 128)
  int iters;             // stages per workgroup
  uint32_t* sink;
};

typedef uint32_t fb_u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void fb_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// WAVES wavefronts; a stage = WAVES * PIECES wave-instructions of 64 * SZ bytes; STAGES ring buffers, STAGES - 1 stages in flight
template <int WAVES, int STAGES, int PIECES, int PATTERN, int SZ, int AUX, bool BARRIER, bool READS>
__global__ __launch_bounds__(64 * WAVES) void k_fill(const FbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PIECE_B = 64 * SZ;
  constexpr int STAGE_B = WAVES * PIECES * PIECE_B;
  static_assert(PIECES * (STAGES - 1) <= 60, "vmcnt is a 6-bit counter");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const fb_rsrc_t rs = FB_MAKE_RSRC(a.src, a.bytes);
  const uint32_t mask = a.bytes - 1u;
  uint32_t lane_off;                                   // byte offset of this lane inside a wave-instruction's source
  if (PATTERN == 0) lane_off = (uint32_t)lane * SZ;
  else if (PATTERN == 1) lane_off = (uint32_t)(lane >> 3) * a.row_stride + (uint32_t)(lane & 7) * SZ;
  else lane_off = (uint32_t)(lane >> 3) * a.row_stride + (uint32_t)((lane & 7) ^ ((lane >> 3) & 7)) * SZ;
  const uint32_t piece_span = (PATTERN == 0) ? (uint32_t)PIECE_B : 8u * a.row_stride;    // source bytes one wave-instruction spans

  auto issue = [&](int it, int buf) {
    // every workgroup walks its own pseudo-random sequence of stage windows through the working set (wave-uniform arithmetic)
    const uint32_t w0 = ((uint32_t)blockIdx.x * 7919u + (uint32_t)it * 104729u) * (uint32_t)(WAVES * PIECES);
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const uint32_t base = ((w0 + (uint32_t)(wave * PIECES + p)) * piece_span) & mask;   // scalar
      const uint32_t voff = (base + lane_off) & mask;
      unsigned char* dst = smem + buf * STAGE_B + (wave * PIECES + p) * PIECE_B;
      // (the builtin wants literal size / policy operands, not value-dependent template arguments)
      if constexpr (SZ == 16 && AUX == 0) FB_LOAD_LDS(rs, dst, 16, voff, 0, 0);
      else if constexpr (SZ == 16 && AUX == 2) FB_LOAD_LDS(rs, dst, 16, voff, 0, 2);
      else if constexpr (SZ == 4 && AUX == 0) FB_LOAD_LDS(rs, dst, 4, voff, 0, 0);
      else static_assert(SZ == 16 && AUX == 0, "add the literal form of this (size, policy) pair");
    }
  };

  uint32_t acc = 0;
  int ibuf = 0;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) {
    if (s < a.iters) issue(s, ibuf);
    ibuf = (ibuf + 1 == STAGES) ? 0 : ibuf + 1;
  }
  int cbuf = 0;
  for (int it = 0; it < a.iters; ++it) {
    const bool more = it + STAGES - 1 < a.iters;
    if (more) {
      issue(it + STAGES - 1, ibuf);
      ibuf = (ibuf + 1 == STAGES) ? 0 : ibuf + 1;
      fb_wait_vm<(STAGES - 1) * PIECES>();             // the oldest stage in flight has landed (this wavefront's pieces)
    } else {
      fb_wait_vm<0>();
    }
    if (BARRIER) __builtin_amdgcn_s_barrier();
    if (READS) {
      // what a 64 x 64 wave tile reads per 64-channel k-step: 16 x 16 bytes per lane (lane-linear rows: conflict-free)
      const uint32_t p = (uint32_t)(uintptr_t)(smem + cbuf * STAGE_B) + (uint32_t)lane * 16u;
      fb_u32x4 v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t q = p + (uint32_t)((r * 1024) % STAGE_B);
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[r]) : "v"(q) : "memory");
      }
      // the wait names every destination register: their uses below cannot be scheduled in front of it
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                     "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                   :
                   : "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) acc ^= v[r][0];
    }
    cbuf = (cbuf + 1 == STAGES) ? 0 : cbuf + 1;
  }
  if (a.iters < 0 || acc == 0x9e3779b9u) a.sink[threadIdx.x] = acc + smem[lane];    // (practically) never executes: keeps the reads alive
}

// The same stage stream WITHOUT LDS-DMA: 16-byte global loads into registers, ds_write_b128 into the ring one stage later (two
// register sets: the loads of stage it + 1 are in flight while stage it is written).  The guide prices an LDS-DMA piece at 60-185
// issue cycles; if that, not the L2, is what bounds the fill, this path (ds_write_b128: ~79 B/clk/CU) is the faster one.
template <int WAVES, int PIECES, int PATTERN, bool BARRIER>
__global__ __launch_bounds__(64 * WAVES) void k_fill_reg(const FbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PIECE_B = 64 * 16;
  constexpr int STAGE_B = WAVES * PIECES * PIECE_B;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t mask = a.bytes - 1u;
  uint32_t lane_off;
  if (PATTERN == 0) lane_off = (uint32_t)lane * 16u;
  else if (PATTERN == 1) lane_off = (uint32_t)(lane >> 3) * a.row_stride + (uint32_t)(lane & 7) * 16u;
  else lane_off = (uint32_t)(lane >> 3) * a.row_stride + (uint32_t)((lane & 7) ^ ((lane >> 3) & 7)) * 16u;
  const uint32_t piece_span = (PATTERN == 0) ? (uint32_t)PIECE_B : 8u * a.row_stride;
  auto fetch = [&](int it, uint4 (&r)[PIECES]) {
    const uint32_t w0 = ((uint32_t)blockIdx.x * 7919u + (uint32_t)it * 104729u) * (uint32_t)(WAVES * PIECES);
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const uint32_t base = ((w0 + (uint32_t)(wave * PIECES + p)) * piece_span) & mask;
      r[p] = *reinterpret_cast<const uint4*>(a.src + ((base + lane_off) & mask));
    }
  };
  // two register sets, the loop unrolled by two by hand: set B's loads are in flight while set A is written and vice versa (a
  // "cur = nxt" copy makes hipcc wait for every load right behind its issue)
  auto put = [&](const uint4 (&r)[PIECES], int buf) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      *reinterpret_cast<uint4*>(smem + buf * STAGE_B + (wave * PIECES + p) * PIECE_B + lane * 16) = r[p];
    if (BARRIER) __syncthreads();
  };
  uint4 ra[PIECES], rb[PIECES];
  // (iters is even and >= 2 -- the host sees to it: no conditional fetch, or the merge of "loaded" and "kept" values costs copies
  // that wait for the loads)
  fetch(0, ra);
  int it = 0;
  for (; it + 2 < a.iters; it += 2) {
    fetch(it + 1, rb);
    put(ra, 0);
    fetch(it + 2, ra);
    put(rb, 1);
  }
  fetch(it + 1, rb);
  put(ra, 0);
  put(rb, 1);
  if (a.iters < 0) a.sink[threadIdx.x] = smem[lane];
}

// Both transports at once: PD LDS-DMA pieces into the ring AND PR 16-byte global loads into registers per wavefront and stage (what a
// kernel does that keeps the pixel operand in LDS and fetches the kernel operand's fragments straight from L2).  Do the two rates
// add up, or do they share one path?  The register loads are inline assembly with counted waits of their own: beside a pending
// LDS-DMA hipcc drains vmcnt(0) for every register-destination load it knows about.
template <int WAVES, int PD, int PR, bool BARRIER, bool RCONT = false>
__global__ __launch_bounds__(64 * WAVES) void k_fill_mix(const FbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PIECE_B = 1024;
  constexpr int STAGE_B = WAVES * PD * PIECE_B;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const fb_rsrc_t rs = FB_MAKE_RSRC(a.src, a.bytes);
  const uint32_t mask = a.bytes - 1u;
  const uint32_t lane_off = (uint32_t)(lane >> 3) * a.row_stride + (uint32_t)((lane & 7) ^ ((lane >> 3) & 7)) * 16u;
  const uint32_t piece_span = 8u * a.row_stride;
  auto dma = [&](int it, int buf) {
    const uint32_t w0 = ((uint32_t)blockIdx.x * 7919u + (uint32_t)it * 104729u) * (uint32_t)(WAVES * (PD + PR));
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const uint32_t base = ((w0 + (uint32_t)(wave * PD + p)) * piece_span) & mask;
      FB_LOAD_LDS(rs, smem + buf * STAGE_B + (wave * PD + p) * PIECE_B, 16, (base + lane_off) & mask, 0, 0);
    }
  };
  auto fetch = [&](int it, fb_u32x4 (&r)[PR > 0 ? PR : 1]) {
    const uint32_t w0 = ((uint32_t)blockIdx.x * 7919u + (uint32_t)it * 104729u) * (uint32_t)(WAVES * (PD + PR)) + (uint32_t)(WAVES * PD);
#pragma unroll
    for (int p = 0; p < PR; ++p) {
      const uint32_t base = ((w0 + (uint32_t)(wave * PR + p)) * piece_span) & mask;
      // RCONT: the register half fetches one contiguous KiB per wave-instruction (a kernel operand pre-packed in fragment order)
      const unsigned char* ptr = a.src + (RCONT ? (((w0 + (uint32_t)(wave * PR + p)) * 1024u + (uint32_t)lane * 16u) & mask) : ((base + lane_off) & mask));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[p]) : "v"(ptr) : "memory");
    }
  };
  uint32_t acc = 0;
  auto use = [&](fb_u32x4 (&r)[PR > 0 ? PR : 1]) {
#pragma unroll
    for (int p = 0; p < PR; ++p) {
      asm volatile("" : "+v"(r[p]));                   // (the wait in front of this call covers them; keeps the use behind it)
      acc ^= r[p][0];
    }
  };
  constexpr int PRA = PR > 0 ? PR : 1;                // (PR == 0: the arrays stay unused)
  fb_u32x4 ra[PRA], rb[PRA];
  // iters even, >= 2.  In flight behind the counted wait: the NEXT stage's PD pieces + PR loads.
  dma(0, 0); fetch(0, ra);
  int it = 0;
  for (; it + 2 < a.iters; it += 2) {
    dma(it + 1, 1); fetch(it + 1, rb);
    fb_wait_vm<PD + PR>();
    use(ra);
    if (BARRIER) __builtin_amdgcn_s_barrier();
    dma(it + 2, 0); fetch(it + 2, ra);
    fb_wait_vm<PD + PR>();
    use(rb);
    if (BARRIER) __builtin_amdgcn_s_barrier();
  }
  dma(it + 1, 1); fetch(it + 1, rb);
  fb_wait_vm<PD + PR>();
  use(ra);
  fb_wait_vm<0>();
  use(rb);
  if (a.iters < 0 || acc == 0x9e3779b9u) a.sink[threadIdx.x] = acc + smem[lane];
}

struct Cfg { const char* name; int waves, stages, pieces, pattern, sz, aux, barrier, reads; };

template <int WAVES, int STAGES, int PIECES, int PATTERN, int SZ, int AUX, bool BARRIER, bool READS>
static float run_one(const FbArgs& a, int wgs_per_cu, int cus, double* bytes_out) {
  constexpr int STAGE_B = WAVES * PIECES * 64 * SZ;
  size_t lds = (size_t)STAGES * STAGE_B;
  // occupancy is set through the LDS request: pad it so that exactly wgs_per_cu workgroups fit in 160 KiB
  const size_t want = (size_t)(160 * 1024) / (size_t)wgs_per_cu;
  if (lds > want) return -1.f;
  lds = want - 1024;                                   // (a little below the share: the runtime may add its own bytes)
  auto kern = k_fill<WAVES, STAGES, PIECES, PATTERN, SZ, AUX, BARRIER, READS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1.f;
  const int grid = cus * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, 64 * WAVES, lds, 0>>>(a);                // warm-up (also warms the cache level under test)
  hipEventRecord(e0, 0);
  kern<<<grid, 64 * WAVES, lds, 0>>>(a);
  hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  *bytes_out = (double)grid * a.iters * STAGE_B;
  return ms;
}

template <int WAVES, int PIECES, int PATTERN, bool BARRIER>
static float run_reg(const FbArgs& a, int wgs_per_cu, int cus, double* bytes_out) {
  constexpr int STAGE_B = WAVES * PIECES * 64 * 16;
  const size_t want = (size_t)(160 * 1024) / (size_t)wgs_per_cu;
  if ((size_t)2 * STAGE_B > want) return -1.f;
  const size_t lds = want - 1024;
  auto kern = k_fill_reg<WAVES, PIECES, PATTERN, BARRIER>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1.f;
  const int grid = cus * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, 64 * WAVES, lds, 0>>>(a);
  hipEventRecord(e0, 0);
  kern<<<grid, 64 * WAVES, lds, 0>>>(a);
  hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  *bytes_out = (double)grid * a.iters * STAGE_B;
  return ms;
}

template <int WAVES, int PD, int PR, bool BARRIER, bool RCONT = false>
static float run_mix(const FbArgs& a, int wgs_per_cu, int cus, double* bytes_out) {
  constexpr int STAGE_B = WAVES * PD * 1024;
  const size_t want = (size_t)(160 * 1024) / (size_t)wgs_per_cu;
  if ((size_t)2 * STAGE_B > want) return -1.f;
  const size_t lds = want - 1024;
  auto kern = k_fill_mix<WAVES, PD, PR, BARRIER, RCONT>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1.f;
  const int grid = cus * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, 64 * WAVES, lds, 0>>>(a);
  hipEventRecord(e0, 0);
  kern<<<grid, 64 * WAVES, lds, 0>>>(a);
  hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  *bytes_out = (double)grid * a.iters * (double)(WAVES * (PD + PR) * 1024);
  return ms;
}

#define FB_CASE(W, S, P, PAT, SZ, AUX, BAR, RD)                                                                          \
  if (c.waves == W && c.stages == S && c.pieces == P && c.pattern == PAT && c.sz == SZ && c.aux == AUX && c.barrier == BAR && c.reads == RD) \
    return run_one<W, S, P, PAT, SZ, AUX, (BAR != 0), (RD != 0)>(a, wgs_per_cu, cus, bytes_out);

static float dispatch(const Cfg& c, const FbArgs& a, int wgs_per_cu, int cus, double* bytes_out) {
  // the product's shape: 4 wavefronts, 8 KiB-pieces per wavefront and stage (32 KiB stages), 2 stages, barrier
  FB_CASE(4, 2, 8, 0, 16, 0, 1, 0) FB_CASE(4, 2, 8, 1, 16, 0, 1, 0) FB_CASE(4, 2, 8, 2, 16, 0, 1, 0)
  FB_CASE(4, 2, 8, 2, 16, 2, 1, 0) FB_CASE(4, 2, 8, 2, 16, 0, 0, 0) FB_CASE(4, 2, 8, 2, 16, 0, 1, 1)
  FB_CASE(4, 3, 8, 2, 16, 0, 1, 0) FB_CASE(4, 3, 8, 2, 16, 0, 0, 0)
  FB_CASE(8, 2, 4, 2, 16, 0, 1, 0) FB_CASE(8, 3, 4, 2, 16, 0, 1, 0) FB_CASE(8, 3, 4, 0, 16, 0, 0, 0)
  FB_CASE(4, 2, 32, 0, 4, 0, 1, 0) FB_CASE(4, 2, 32, 0, 4, 0, 0, 0)
  FB_CASE(4, 2, 4, 2, 16, 0, 1, 0) FB_CASE(4, 4, 4, 2, 16, 0, 1, 0) FB_CASE(4, 4, 4, 2, 16, 0, 0, 0)
  if (c.stages == -1) {                                // mixed transport: pieces = LDS-DMA pieces, sz = register loads per wavefront and stage
    if (c.pieces == 4 && c.sz == 4 && c.barrier == 1 && c.aux == 0) return run_mix<4, 4, 4, true>(a, wgs_per_cu, cus, bytes_out);
    if (c.pieces == 4 && c.sz == 4 && c.barrier == 0) return run_mix<4, 4, 4, false>(a, wgs_per_cu, cus, bytes_out);
    if (c.pieces == 8 && c.sz == 8 && c.barrier == 1) return run_mix<4, 8, 8, true>(a, wgs_per_cu, cus, bytes_out);
    if (c.pieces == 4 && c.sz == 0 && c.barrier == 1) return run_mix<4, 4, 0, true>(a, wgs_per_cu, cus, bytes_out);
    if (c.pieces == 0 && c.sz == 4 && c.barrier == 1 && c.aux == 0) return run_mix<4, 0, 4, true>(a, wgs_per_cu, cus, bytes_out);
    if (c.pieces == 4 && c.sz == 4 && c.barrier == 1 && c.aux == 1) return run_mix<4, 4, 4, true, true>(a, wgs_per_cu, cus, bytes_out);
    if (c.pieces == 4 && c.sz == 8 && c.barrier == 1 && c.aux == 1) return run_mix<4, 4, 8, true, true>(a, wgs_per_cu, cus, bytes_out);
    if (c.pieces == 0 && c.sz == 8 && c.barrier == 1 && c.aux == 1) return run_mix<4, 0, 8, true, true>(a, wgs_per_cu, cus, bytes_out);
  }
  if (c.stages == 0) {                                 // register-staged: (waves, pieces, pattern, barrier)
    if (c.waves == 4 && c.pieces == 8 && c.pattern == 0 && c.barrier == 1) return run_reg<4, 8, 0, true>(a, wgs_per_cu, cus, bytes_out);
    if (c.waves == 4 && c.pieces == 8 && c.pattern == 2 && c.barrier == 1) return run_reg<4, 8, 2, true>(a, wgs_per_cu, cus, bytes_out);
    if (c.waves == 4 && c.pieces == 8 && c.pattern == 2 && c.barrier == 0) return run_reg<4, 8, 2, false>(a, wgs_per_cu, cus, bytes_out);
    if (c.waves == 8 && c.pieces == 4 && c.pattern == 2 && c.barrier == 1) return run_reg<8, 4, 2, true>(a, wgs_per_cu, cus, bytes_out);
  }
  return -2.f;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  printf("# %s: %d CUs, %.2f GHz (clockRate), L2 %d KiB\n", prop.name, cus, ghz, prop.l2CacheSize / 1024);
  const size_t max_bytes = (size_t)1 << 30;
  unsigned char* src = nullptr;
  if (hipMalloc(&src, max_bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemset(src, 1, max_bytes);
  uint32_t* sink = nullptr;
  hipMalloc(&sink, 4096);
  const Cfg cfgs[] = {
      {"contiguous KiB, 4w 2st 32K, barrier", 4, 2, 8, 0, 16, 0, 1, 0},
      {"row gather,     4w 2st 32K, barrier", 4, 2, 8, 1, 16, 0, 1, 0},
      {"gather+swizzle, 4w 2st 32K, barrier   (the product)", 4, 2, 8, 2, 16, 0, 1, 0},
      {"gather+swizzle, 4w 2st 32K, barrier, nt", 4, 2, 8, 2, 16, 2, 1, 0},
      {"gather+swizzle, 4w 2st 32K, no barrier", 4, 2, 8, 2, 16, 0, 0, 0},
      {"gather+swizzle, 4w 2st 32K, barrier, + fragment reads", 4, 2, 8, 2, 16, 0, 1, 1},
      {"gather+swizzle, 4w 3st 32K, barrier", 4, 3, 8, 2, 16, 0, 1, 0},
      {"gather+swizzle, 4w 3st 32K, no barrier", 4, 3, 8, 2, 16, 0, 0, 0},
      {"gather+swizzle, 8w 2st 32K, barrier", 8, 2, 4, 2, 16, 0, 1, 0},
      {"gather+swizzle, 8w 3st 32K, barrier", 8, 3, 4, 2, 16, 0, 1, 0},
      {"contiguous KiB, 8w 3st 32K, no barrier", 8, 3, 4, 0, 16, 0, 0, 0},
      {"contiguous 256 B (4 B / lane), 4w 2st 32K, barrier", 4, 2, 32, 0, 4, 0, 1, 0},
      {"contiguous 256 B (4 B / lane), 4w 2st 32K, no barrier", 4, 2, 32, 0, 4, 0, 0, 0},
      {"gather+swizzle, 4w 2st 16K, barrier", 4, 2, 4, 2, 16, 0, 1, 0},
      {"gather+swizzle, 4w 4st 16K, barrier", 4, 4, 4, 2, 16, 0, 1, 0},
      {"gather+swizzle, 4w 4st 16K, no barrier", 4, 4, 4, 2, 16, 0, 0, 0},
      {"REGISTER-staged (global_load -> ds_write_b128), contiguous KiB, 4w 32K, barrier", 4, 0, 8, 0, 16, 0, 1, 0},
      {"REGISTER-staged, gather+swizzle, 4w 32K, barrier", 4, 0, 8, 2, 16, 0, 1, 0},
      {"REGISTER-staged, gather+swizzle, 4w 32K, no barrier", 4, 0, 8, 2, 16, 0, 0, 0},
      {"REGISTER-staged, gather+swizzle, 8w 32K, barrier", 8, 0, 4, 2, 16, 0, 1, 0},
      {"MIXED: 4 LDS-DMA pieces + 4 register loads per wave and stage (16K + 16K), barrier", 4, -1, 4, 2, 4, 0, 1, 0},
      {"MIXED: 4 LDS-DMA pieces + 4 register loads per wave and stage, no barrier", 4, -1, 4, 2, 4, 0, 0, 0},
      {"MIXED: 8 + 8 per wave and stage (32K + 32K), barrier", 4, -1, 8, 2, 8, 0, 1, 0},
      {"MIXED kernel, LDS-DMA half only (4 pieces, 16K), barrier", 4, -1, 4, 2, 0, 0, 1, 0},
      {"MIXED kernel, register half only (4 loads, 16K, nothing written to LDS), barrier", 4, -1, 0, 2, 4, 0, 1, 0},
      {"MIXED: 4 LDS-DMA pieces + 4 CONTIGUOUS-KiB register loads per wave and stage, barrier", 4, -1, 4, 2, 4, 1, 1, 0},
      {"MIXED: 4 LDS-DMA pieces + 8 CONTIGUOUS-KiB register loads per wave and stage, barrier", 4, -1, 4, 2, 8, 1, 1, 0},
      {"MIXED kernel, 8 CONTIGUOUS-KiB register loads only, barrier", 4, -1, 0, 2, 8, 1, 1, 0},
  };
  const size_t sets[] = {(size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)1 << 30};
  const uint32_t strides[] = {256u, 2048u};            // 128-channel and 1024-channel bf16 rows
  printf("%-82s %5s %7s %8s | %9s %8s\n", "configuration", "WG/CU", "set MiB", "stride", "GB/s", "B/clk/CU");
  for (const Cfg& c : cfgs) {
    for (int wgs = 1; wgs <= 2; ++wgs) {
      for (size_t set : sets) {
        for (uint32_t stride : strides) {
          if (c.pattern == 0 && stride != strides[0]) continue;
          FbArgs a;
          a.src = src; a.bytes = (uint32_t)set; a.row_stride = stride; a.sink = sink;
          const int stage_b = (c.stages == -1) ? c.waves * (c.pieces + c.sz) * 1024 : c.waves * c.pieces * 64 * c.sz;
          a.iters = (int)(((size_t)96 << 20) / (size_t)stage_b);        // 96 MiB per workgroup
          if (a.iters > 4096) a.iters = 4096;
          a.iters &= ~1;                                                   // (k_fill_reg: an even count)
          double bytes = 0.0;
          const float ms = dispatch(c, a, wgs, cus, &bytes);
          if (ms <= 0.f) continue;
          const double gbs = bytes / (ms * 1e-3) * 1e-9;
          printf("%-82s %5d %7zu %8u | %9.0f %8.1f\n", c.name, wgs, set >> 20, stride, gbs, gbs / (cus * ghz));
          fflush(stdout);
        }
      }
    }
  }
  hipFree(src); hipFree(sink);
  return 0;
}
