// Shared by the implicit-GEMM kernels (pf_igemm.hip): argument block,
// buffer-descriptor / LDS-DMA macros, counted waits, mode constants.
#pragma once
#include "pf_conv_common.h"

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// The buffer-descriptor type exists only in the device pass; the host pass (which merely emits the launch stub) still has
// to parse the kernel body -- without this the host pass silently DROPS the stubs and the library fails to load.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t pf_rsrc_t;
#define PF_MAKE_RSRC(p, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(p), (short)0, (int)(bytes), 0x00020000)
#define PF_BUFFER_LOAD_LDS16(rs, lds, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, voff, soff, 0, 0)
#else
typedef int pf_rsrc_t;
#define PF_MAKE_RSRC(p, bytes) 0
#define PF_BUFFER_LOAD_LDS16(rs, lds, voff, soff) ((void)(rs), (void)(lds), (void)(voff), (void)(soff))
#endif

struct IgArgs {
  const bf16_t* X;      // [rows_in][C]
  const bf16_t* W;      // [N][taps][C]
  bf16_t* Y;            // [M][N]
  const bf16_t* zero;   // unused by the kernel (padding taps read zeros through the buffer bounds check); kept in the ABI
  uint32_t x_bytes, w_bytes;   // sizes of X / W in bytes (buffer descriptors)
  const bf16_t* R;      // residual [M][N] or null
  float* partial;       // statistics [G][4][N] (or [G][2][N] with bx) or null
  const bf16_t* bx;     // BN-backward statistics mode: the BN's input x [M][N]
  const float* bss;     // its scale | shift [2][N]
  const float* bmi;     // its mean | invstd [2][N]
  float b_lo, b_hi;
  const float* ss;      // PRO: scale | shift [2][C] of the producer BN
  const uint32_t* slot; // PRO: activation range (null: no fake-quant)
  float kq, act_lo, act_hi;
  int M, N, C;
  int th, tw;           // taps
  int H, Wd, Ho, Wo, stride, pad_h, pad_w;
  int tiles_m, tiles_n, G;
  // sub-filter walk (strided backward-data by output-parity classes, pf_conv2d_bwd_data_strided): the th x tw taps of THIS launch are
  // taps (w_r0 + r * w_rs, w_s0 + s * w_ss) of a kernel buffer whose rows hold w_taps_full taps, w_S of them per kernel row.
  // A plain convolution walks its own kernel: w_r0 = w_s0 = 0, w_rs = w_ss = 1, w_S = tw, w_taps_full = th * tw.
  int w_r0, w_rs, w_s0, w_ss, w_S, w_taps_full;
  // output scatter: row (img, i, j) of the launch's [Ho x Wo] grid is stored at pixel (i * o_sub + o_y, j * o_sub + o_x) of an
  // [o_H x o_W] image (o_sub = 0: off, rows are stored where they are)
  int o_sub, o_y, o_x, o_H, o_W;
  // output affine of the row pass (pf_conv_common.h out_affine8): the consumer's inference-mode BN + activation; null: off
  const float* oss;     // scale | shift [2][N]
  int oact;
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: plain (+ residual / statistics), 1: backward-data with BN-backward sums, 2: producer's BN + act + fake-quant
// prologue on the input operand (1x1 only: padding taps would need Q = 0, not Q(0))
#define IG_PLAIN 0
#define IG_BWD 1
#define IG_PRO 2
