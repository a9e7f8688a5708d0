// tools/gpu/simd_probe.hip -- which SIMD does wavefront w of a 512-thread workgroup run on?  (tool only)
// The ping-pong kernel (pocketflow_amd/csrc/pf_igemm_pp.hip) puts wavefronts 0-3 and 4-7 into two groups and relies on wavefronts
// w and w + 4 sharing a SIMD.  Prints HW_ID's SIMD_ID field per wavefront for a few workgroups.
//   build + run:  hipcc --offload-arch=gfx950 -O2 -o simd_probe simd_probe.hip && ./simd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(512) void k_probe(uint32_t* out) {
  extern __shared__ unsigned char smem[];
  uint32_t hw;
  // s_getreg_b32 hwreg(HW_REG_HW_ID = 4): SIMD_ID = bits [5:4], WAVE_ID [3:0], CU_ID [11:8], SE_ID [15:13]
  asm volatile("s_getreg_b32 %0, hwreg(4, 0, 32)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
  if (threadIdx.x == 9999) smem[0] = 1;
}

int main() {
  const int blocks = 512;
  uint32_t* d;
  hipMalloc(&d, blocks * 8 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  k_probe<<<blocks, 512, 144 * 1024, 0>>>(d);   // LDS request as the ping-pong kernel's: one workgroup per CU
  uint32_t h[blocks * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int pairs_ok = 0, total = 0;
  for (int b = 0; b < blocks; ++b) {
    bool ok = true;
    for (int w = 0; w < 4; ++w) ok = ok && (((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3));
    pairs_ok += ok; ++total;
    if (b < 6 || !ok) {
      printf("block %3d: SIMD of wavefronts 0..7 =", b);
      for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3);
      printf("   (cu %u se %u)%s\n", (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7, ok ? "" : "   <-- w and w+4 do NOT share a SIMD");
      if (!ok && b > 40) break;
    }
  }
  printf("workgroups in which wavefronts w and w + 4 share a SIMD for every w: %d of %d\n", pairs_ok, total);
  return 0;
}
