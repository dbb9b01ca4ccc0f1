// Shared by the two implicit-GEMM kernels (conv_gemm.hip: everything through LDS, any layout / precision;
// conv_gemm_ad.hip: the f16x3 fast path with activations loaded straight into MFMA fragments).
#pragma once
#include "common.h"

namespace mspi {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));

// Precision modes.
//  PREC_F32   : v_mfma_f32_32x32x2_f32 on fp32 operands (exact fp32 fmaf chain).
//  PREC_F16X3 : fp32-accurate product on the 16x faster f16 matrix pipe.  Every operand is split
//               x = hi + lo with hi = f16(x), lo = f16(x - hi) (22 significand bits together) and
//               the product is accumulated in fp32 as hi*hi + hi*lo + lo*hi by three
//               v_mfma_f32_32x32x16_f16 (the dropped lo*lo term is 2^-22 relative; f16xf16
//               products are exact in fp32).  Weights are split once at pack time (pre-scaled by a
//               power of two so their lo part stays a normal f16); activations are split while they
//               are staged into LDS.
enum { PREC_F32 = 0, PREC_F16X3 = 1 };

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  const float* gate;
  float* y;
  int N, T, H, W, C;
  long sN, sT, sH, sW, sC;
  int kT, kH, kW, strT, strH, strW, padT, padH, padW;
  int To, Ho, Wo, Cout;
  long ldy, ldw, ldr;
  int act;
  int M, K;
  int rows_per_sample;
  int tiles_n, nblocks;
  float out_scale;  // F16X3: 1 / (power-of-two weight pre-scale), applied to the accumulator
  int* status;      // range guard (common.h): set to 1 when an output is not finite; may be NULL
  int dbg;          // ablation switches for tools/gemm_probe.py (MSPI_CONV_DBG); 0 in production
  int ksplit;       // split-K: gridDim.y workgroups share an output tile, each owns a contiguous range of K steps ...
  float* ws;        // ... and writes its partial sums to ws[z][M][Cout] (no bias / residual / activation); NULL: no split
  // pre-split activations (conv_gemm_dma_kernel<..., APRE>): A is two BLOCKED f16 planes (common.h plane_off; ldxs == K) (hi, then lo `xplane` elements
  // later), written by a producer's epilogue; the kernel then DMAs them like the weight planes and splits nothing
  int dense_rows;       // 1: 1x1x1 / stride 1 / no padding on an input whose (n,t,h,w) positions are equally spaced rows (offset = m * sW):
                        // the loaders skip the per-row integer divisions of the general gather
  const _Float16* wb;   // blocked f16 hi/lo weight planes (common.h plane_off; rows padded to 16), or NULL: the LDS-DMA kernels' weight source
  const _Float16* xs;
  long ldxs, xplane;
  // split-plane OUTPUT (any kernel's epilogue): when ys != NULL the result goes to two f16 planes instead of y
  _Float16* ys;
  long ldys, yplane;
};

constexpr int BK = 32;

// XCD-aware, bijective block remap: blocks dealt to one XCD (bid % 8) get consecutive logical ids, so the
// N-tiles that re-read one A row panel (and the M-tiles that re-read one weight panel) share that XCD's L2.
__device__ __forceinline__ int xcd_logical_block(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

}  // namespace mspi
