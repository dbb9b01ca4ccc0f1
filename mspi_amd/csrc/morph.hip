// MorphMLP data movement (backbones/MorphMLP.py:38-158): the MorphFC layers are plain Linears applied to tokens that have
// been regrouped -- `reshape / permute / reshape` chains upstream, each of which materialises a copy.  Here a regrouping is
// ONE strided-gather launch (any permutation of up to 6 dims whose innermost run is contiguous on both sides), and the
// "reweight" mix  h*a0 + w*a1 + c*a2  with a = softmax over the branches of the re-weighting MLP's logits is one
// element-wise launch.  Both are HBM-bound: 8 bytes per element moved.
#include "common.h"

namespace mspi {

struct PermArgs {
  const float* x;
  float* y;
  int d[6];        // output extents, d[5] innermost (in units of VEC floats)
  long s[6];       // input strides (floats) of the same six indices
  long total;      // output elements / VEC
};

template <int VEC>
__global__ __launch_bounds__(256) void permute_kernel(const PermArgs p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  long r = idx;
  long off = 0;
#pragma unroll
  for (int k = 5; k >= 1; --k) {
    const long q = r / p.d[k];
    const int i = (int)(r - q * p.d[k]);
    off += (long)i * p.s[k] * (k == 5 ? VEC : 1);
    r = q;
  }
  off += r * p.s[0];
  if (VEC == 4) {
    *reinterpret_cast<float4*>(p.y + idx * 4) = *reinterpret_cast<const float4*>(p.x + off);
  } else {
    p.y[idx] = p.x[off];
  }
}

// y[n,r,c] = sum_j softmax_j(logit[n, c*J + j]) * src_j[n,r,c]   (MorphFC_S :104-107, MorphFC_S2 :64-67)
template <int J>
__global__ __launch_bounds__(256) void gated_sum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ c, const float* __restrict__ logit,
                                                        float* __restrict__ y, long rows_per_sample, int C, long total4) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int CV = C >> 2;
  const int cv = (int)(idx % CV);
  const long row = idx / CV;
  const long n = row / rows_per_sample;
  const float* lg = logit + (n * C + cv * 4) * J;
  const float4 va = *reinterpret_cast<const float4*>(a + idx * 4);
  const float4 vb = *reinterpret_cast<const float4*>(b + idx * 4);
  float4 vc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (J == 3) vc = *reinterpret_cast<const float4*>(c + idx * 4);
  const float xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w}, xc[4] = {vc.x, vc.y, vc.z, vc.w};
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float l0 = lg[k * J], l1 = lg[k * J + 1], l2 = J == 3 ? lg[k * J + 2] : -INFINITY;
    const float m = fmaxf(fmaxf(l0, l1), l2);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = J == 3 ? expf(l2 - m) : 0.f;
    const float inv = 1.f / (e0 + e1 + e2);
    o[k] = (e0 * inv) * xa[k] + (e1 * inv) * xb[k] + (J == 3 ? (e2 * inv) * xc[k] : 0.f);
  }
  *reinterpret_cast<float4*>(y + idx * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace mspi

using namespace mspi;

extern "C" int mspi_permute_fwd(const MspiPermuteDesc* d, const float* x, float* y, mspi_stream_t stream) {
  MSPI_REQUIRE(d && x && y, "mspi_permute_fwd: null argument");
  PermArgs a;
  a.x = x; a.y = y;
  long total = 1, span = 0;
  for (int k = 0; k < 6; ++k) {
    MSPI_REQUIRE(d->dims[k] > 0 && d->strides[k] >= 0, "mspi_permute_fwd: dims[%d]=%d strides[%d]=%ld", k, d->dims[k], k,
                 (long)d->strides[k]);
    a.d[k] = d->dims[k]; a.s[k] = d->strides[k];
    total *= d->dims[k];
    span += (long)(d->dims[k] - 1) * d->strides[k];
  }
  // the gather must stay inside the source buffer the caller declares
  MSPI_REQUIRE(span < d->src_elems, "mspi_permute_fwd: strides reach element %ld of a %ld-element source", span,
               (long)d->src_elems);
  MSPI_REQUIRE(d->strides[5] == 1, "mspi_permute_fwd: the innermost run must be contiguous (stride 1)");
  bool v4 = (d->dims[5] & 3) == 0 && aligned16(x) && aligned16(y);
  for (int k = 0; k < 5; ++k) v4 = v4 && (d->strides[k] & 3) == 0;
  if (v4) { a.d[5] >>= 2; total >>= 2; }
  a.total = total;
  MSPI_REQUIRE((total + 255) / 256 < (1L << 31), "mspi_permute_fwd: grid too large");
  const dim3 grid((unsigned)((total + 255) / 256));
  if (v4) hipLaunchKernelGGL(permute_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(permute_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("mspi_permute_fwd");
}

extern "C" int mspi_gated_sum_fwd(const float* a, const float* b, const float* c, const float* logit, float* y, int32_t N,
                                  int64_t rows_per_sample, int32_t C, int32_t J, mspi_stream_t stream) {
  MSPI_REQUIRE(a && b && logit && y && (J == 2 || (J == 3 && c)), "mspi_gated_sum_fwd: null argument or J not in {2,3}");
  MSPI_REQUIRE(N > 0 && rows_per_sample > 0 && C > 0 && (C & 3) == 0, "mspi_gated_sum_fwd: C must be a positive multiple of 4");
  MSPI_REQUIRE(aligned16(a) && aligned16(b) && (!c || aligned16(c)) && aligned16(y), "mspi_gated_sum_fwd: pointers must be 16-B aligned");
  const long total4 = (long)N * rows_per_sample * (C / 4);
  MSPI_REQUIRE((total4 + 255) / 256 < (1L << 31), "mspi_gated_sum_fwd: grid too large");
  const dim3 grid((unsigned)((total4 + 255) / 256));
  if (J == 3) hipLaunchKernelGGL(gated_sum_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a, b, c, logit, y, (long)rows_per_sample, C, total4);
  else hipLaunchKernelGGL(gated_sum_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a, b, c, logit, y, (long)rows_per_sample, C, total4);
  return check_launch("mspi_gated_sum_fwd");
}
