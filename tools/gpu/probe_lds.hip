// Hardware-semantics probe (run once on an MI355X; results recorded in DESIGN.md):
//  1. ds_read_b64_tr_b16: which element each lane receives when the 16 lanes of a group point at the 16 8-byte pieces of a
//     [4 rows][16 columns] bf16 block (row stride 32 / 64 bytes);
//  2. global_load_lds (16 bytes per lane): per-lane GLOBAL addresses, LDS destination = wave-uniform base + lane * 16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__global__ void k_tr(uint16_t* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x, grp = lane >> 4, i = lane & 15;
  const int row = i >> 2, piece = i & 3;
  const uint16_t* p = lds + grp * 512 + (row * stride_bytes) / 2 + piece * 4;
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(v4s, p));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}

__global__ void k_glds(const uint32_t* g, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t* src = g + ((lane ^ 5) + wave * 64) * 4;                  // per-lane source: permuted 16-byte pieces
  __builtin_amdgcn_global_load_lds(src, LDS_PTR(void, lds + wave * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = lds[i];
}

int main() {
  uint16_t* d; (void)hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {32, 64}) {
    k_tr<<<1, 64>>>(d, stride);
    uint16_t h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16, row stride %d bytes (lds[i] = i; group g block at 512*g; lane i -> piece (row i/4, cols 4*(i%%4)..))\n", stride);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l & 3) == 3 ? "\n" : "   ");
  }
  uint32_t hg[512], *dg, *dout;
  for (int i = 0; i < 512; ++i) hg[i] = i;
  (void)hipMalloc(&dg, sizeof(hg)); (void)hipMalloc(&dout, sizeof(hg));
  (void)hipMemcpy(dg, hg, sizeof(hg), hipMemcpyHostToDevice);
  k_glds<<<1, 128>>>(dg, dout);
  uint32_t ho[512]; (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 2; ++w) for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j)
    if (ho[w * 256 + l * 4 + j] != (uint32_t)(((l ^ 5) + w * 64) * 4 + j)) ++bad;
  printf("global_load_lds: lds[wave*256 + lane*4 + j] == g[((lane^5) + wave*64)*4 + j] : %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %u %u %u %u\n", l, ho[l*4], ho[l*4+1], ho[l*4+2], ho[l*4+3]);
  return 0;
}
