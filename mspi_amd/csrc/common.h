// Shared host/device helpers for libmspi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/mspi_hip.h"
#include <type_traits>

namespace mspi {

void set_error(const char* fmt, ...);

// Check a launch; HIP launch errors surface through hipGetLastError (no sync: capture safe).
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MSPI_ELAUNCH;
  }
  return MSPI_OK;
}

#define MSPI_REQUIRE(cond, ...)   \
  do {                            \
    if (!(cond)) {                \
      mspi::set_error(__VA_ARGS__); \
      return MSPI_EINVAL;         \
    }                             \
  } while (0)

// -DMSPI_F16_SINGLE_PRODUCT (make SINGLE=1 -> libmspi_hip_single.so, selected with MSPI_LIB_PATH): measurement build in which
// every f16x3 kernel issues only the hi*hi MFMA of its three split products, i.e. plain f16 operands with fp32 accumulation
// (BASELINE configs[4] "fp16 MFMA").  Same data path, a third of the MFMAs.  DESIGN.md section 5 records what that costs in
// accuracy on the golden vectors.  A COMPILE-time constant on purpose: a run-time flag puts a branch around two of every
// three MFMAs, and those basic-block boundaries break the MFMA / LDS-read interleaving of the production kernels.
#ifdef MSPI_F16_SINGLE_PRODUCT
constexpr bool kSingleProduct = true;
#else
constexpr bool kSingleProduct = false;
#endif

// Range guard of the f16x3 kernels.  An activation with |x| >= 65504 becomes inf in its f16 hi half and the product sums turn
// into inf / NaN: every GEMM epilogue checks its pre-activation results and, on a non-finite one, stores 1 into the status
// word the caller registered with mspi_set_status_word() -- a word of pinned HOST memory that kernels address directly, so the
// caller reads it without a device synchronisation of its own (after the event that covers the launches).  NULL: no report.
extern int* g_status_word;

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 rounding level): libm's erff inlines to
// ~40 instructions per call site and GEMM epilogues have 64 of them.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));   // v_rcp_f32 (1 ulp); __frcp_rn expands to a full IEEE division
  float y = fmaf(1.061405429f, t, -1.453152027f);
  y = fmaf(y, t, 1.421413741f);
  y = fmaf(y, t, -0.284496736f);
  y = fmaf(y, t, 0.254829592f);
  y = 1.f - y * t * __expf(-ax * ax);
  return copysignf(y, x);
}

// GELU (erf form) u * Phi(u) with the constants of fast_erf folded onto u: q = 0.5 * erfc(|u| / sqrt 2) = t P(t) exp(-u^2 / 2) with
// t = 1 / (1 + (p / sqrt 2) |u|) and the polynomial's coefficients halved; Phi = 1 - q for u >= 0, q below.  15 VALU instructions
// (two of them transcendental) against 17 for 0.5 u (1 + erf(u / sqrt 2)); same absolute error on Phi (<= 0.75e-7).
__device__ __forceinline__ float gelu_erf(float u) {
  const float au = fabsf(u);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, au, 1.f));
  const float e = __builtin_amdgcn_exp2f(u * u * (-0.5f * 1.4426950408889634f));
  float y = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  y = fmaf(y, t, 0.5f * 1.421413741f);
  y = fmaf(y, t, 0.5f * -0.284496736f);
  y = fmaf(y, t, 0.5f * 0.254829592f);
  const float r = u * (y * t * e);          // u q
  return u >= 0.f ? u - r : r;
}

// sigmoid / swish on v_exp_f32 + v_rcp_f32 (1 ulp each): the IEEE `/` expands to a ~10-instruction div_scale / fma / div_fixup
// sequence per element, which was most of the VALU work of the X3D depthwise epilogues and of the gate prologues.
__device__ __forceinline__ float fast_sigmoid(float v) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}
__device__ __forceinline__ float fast_swish(float v) { return v * fast_sigmoid(v); }

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case MSPI_ACT_RELU: return fmaxf(v, 0.f);
    case MSPI_ACT_GELU: return gelu_erf(v);  // nn.GELU (erf form)
    case MSPI_ACT_SIGMOID: return fast_sigmoid(v);
    case MSPI_ACT_SWISH: return fast_swish(v);
    default: return v;
  }
}

// Run `fn(std::integral_constant<int, ACT>())` for the runtime activation code: the element loops inside `fn` then see a
// compile-time ACT.  With `act` looked at per element the compiler keeps the whole switch (scalar compares and branches, the
// sigmoid's IEEE division) around every value: hoisting it out of the GEMM epilogues alone was worth 6 % of the bench line.
#define MSPI_DISPATCH_ACT(act, fn)                                                       \
  switch (act) {                                                                          \
    case MSPI_ACT_RELU: fn(std::integral_constant<int, MSPI_ACT_RELU>()); break;          \
    case MSPI_ACT_GELU: fn(std::integral_constant<int, MSPI_ACT_GELU>()); break;          \
    case MSPI_ACT_SIGMOID: fn(std::integral_constant<int, MSPI_ACT_SIGMOID>()); break;    \
    case MSPI_ACT_SWISH: fn(std::integral_constant<int, MSPI_ACT_SWISH>()); break;        \
    default: fn(std::integral_constant<int, MSPI_ACT_NONE>()); break;                     \
  }

// hi/lo f16 split of an fp32 value: x ~= hi + lo to 22 significand bits (the f16x3 operand form, conv_common.h).
// x is pinned in a register first.  If the compiler may contract the multiply that PRODUCED x into the subtraction
// below (fma(a, b, -hi)), hi is rounded from fl32(a*b) but lo is taken from the unrounded a*b, and in rare
// double-rounding cases the pair is off by one f16 ulp of hi (2^-11): measured as 2e-5 outliers in 0.2 % of the rows
// of the f16x3 attention when q was multiplied by a non-power-of-two scale right before its split.
__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
  asm("" : "+v"(x));
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

// Pre-split activation planes are BLOCKED: 16 rows x 32 k halves = 1 KB contiguous per block, blocks k-fastest, so that the
// k32 stage of a 16-row group is ONE contiguous 1-KB LDS-DMA piece of 8 full cache lines (row-major [M][K] planes hand the
// loader 16 half-lines per piece: 30 instead of 43 B/clk/CU of fill, tools/dma_issue_probe.hip).  kt = K / 32; rows are
// allocated up to a multiple of 16.
__device__ __forceinline__ long plane_off(long m, int k, int kt) {
  return ((m >> 4) * kt + (k >> 5)) * 512 + (m & 15) * 32 + (k & 31);
}

// acc += x * w on four channels as TWO packed fp32 FMAs (v_pk_fma_f32, per-half operands: .xy and .zw of a float4 are aligned
// register pairs).  The 7x7 depthwise kernels are VALU-bound on their 49 FMAs per output; the packed form halves the
// instruction count with the same IEEE fma per element.  Only the PER-HALF form is used: the cross-half op_sel form is the one
// that misbehaves beside MFMA kernels (tools/pk_overlap_probe.hip; tools/check_no_packed_f32.py guards it).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fma4(float4& acc, const float4& x, const float4& w) {
#ifdef MSPI_DW_NO_PK      // A/B build: the scalar form
  acc.x = fmaf(x.x, w.x, acc.x); acc.y = fmaf(x.y, w.y, acc.y); acc.z = fmaf(x.z, w.z, acc.z); acc.w = fmaf(x.w, w.w, acc.w);
  return;
#endif
  v2f a0 = {acc.x, acc.y}, a1 = {acc.z, acc.w};
  const v2f x0 = {x.x, x.y}, x1 = {x.z, x.w}, w0 = {w.x, w.y}, w1 = {w.z, w.w};
  a0 = __builtin_elementwise_fma(x0, w0, a0);
  a1 = __builtin_elementwise_fma(x1, w1, a1);
  acc.x = a0[0]; acc.y = a0[1]; acc.z = a1[0]; acc.w = a1[1];
}

__device__ __forceinline__ bool nonfinite(float v) { return !(fabsf(v) <= 3.4028235e38f); }
__device__ __forceinline__ void report_nonfinite(int* status, bool bad) {
  if (bad && status) *reinterpret_cast<volatile int*>(status) = 1;   // idempotent: racing stores write the same value
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace mspi
