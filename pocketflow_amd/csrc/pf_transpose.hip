// Backward-data layout of every convolution kernel in ONE launch.
// Conv2DBackpropInput contracts over the OUTPUT channels of the kernel: dX = conv(dY, W') with
//   W'[c][R-1-r][S-1-s][n] = W[n][r][s][c]        (KRSC storage of both; a plain transpose for 1x1 kernels)
// The layer executor used to build W' per layer and step with aten (~50 small copy / flip launches, 0.45 ms of a 30 ms
// step); here a tile table built once on the host (one 48-byte descriptor per 64 x 64 tile of every (kernel, tap) matrix)
// drives a single LDS-transposing kernel over the flat compute-dtype buffer.
#include "pf_common.h"

struct PfTile {          // include/pocketflow_hip.h: PfTransposeTile
  int64_t src_off;       // element offset of W[0][rs][0] inside the flat buffer
  int64_t dst_off;       // element offset of W'[0][rs'][0]
  int32_t O, I;          // matrix [O][I] (row stride src_ld) -> [I][O] (row stride dst_ld)
  int32_t src_ld, dst_ld;
  int32_t o0, i0;        // tile origin
  int32_t pad0, pad1;
};

template <typename T>
__global__ __launch_bounds__(PF_THREADS) void k_seg_transpose(const T* __restrict__ src, T* __restrict__ dst,
                                                              const PfTile* __restrict__ tiles) {
  __shared__ T tile[64][65];
  const PfTile t = tiles[blockIdx.x];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;        // 64 x 4
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int o = t.o0 + r, i = t.i0 + tx;
    if (o < t.O && i < t.I) tile[r][tx] = src[t.src_off + (int64_t)o * t.src_ld + i];
  }
  __syncthreads();
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int i = t.i0 + r, o = t.o0 + tx;
    if (o < t.O && i < t.I) dst[t.dst_off + (int64_t)i * t.dst_ld + o] = tile[tx][r];
  }
}

extern "C" int pf_seg_transpose(const void* src_flat, void* dst_flat, int dtype, const void* tiles, int n_tiles,
                                void* stream) {
  if (n_tiles <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PF_F32)
    k_seg_transpose<float><<<n_tiles, PF_THREADS, 0, st>>>((const float*)src_flat, (float*)dst_flat, (const PfTile*)tiles);
  else if (dtype == PF_BF16)
    k_seg_transpose<bf16_t><<<n_tiles, PF_THREADS, 0, st>>>((const bf16_t*)src_flat, (bf16_t*)dst_flat, (const PfTile*)tiles);
  else
    return (int)hipErrorInvalidValue;
  PF_LAUNCH_CHECK();
  return 0;
}
