// Saliency metrics as device reductions: KL divergence, linear correlation (CC), similarity (SIM) and normalised
// scanpath saliency (NSS) of a predicted map against the ground-truth density / fixation map, one workgroup per
// sample, three passes over the H*W values (they stay in L2), fixed-order tree reductions (bitwise reproducible).
// Formulas follow utils/compute_saliency_metrics.py:9-108 of the reference term by term (eps = 2.2204e-16, unbiased
// std, min-max normalisation before SIM); the per-sample values are written, the batch mean is the host's.
#include "common.h"

namespace mspi {

constexpr int MT = 1024;   // threads per workgroup

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();                       // sh may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < MT / 64; ++i) t += sh[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = sh[0];
#pragma unroll
  for (int i = 1; i < MT / 64; ++i) t = fmaxf(t, sh[i]);
  return t;
}

__global__ __launch_bounds__(MT) void saliency_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                              const float* __restrict__ fix, float* __restrict__ out,
                                                              int L, int pred_is_log) {
  __shared__ float sh[MT / 64];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* s_ = pred + (long)n * L;
  const float* g_ = gt + (long)n * L;
  const float* f_ = fix ? fix + (long)n * L : nullptr;
  const float eps = 2.2204e-16f;
  auto S = [&](int i) { return pred_is_log ? __expf(s_[i]) : s_[i]; };

  // pass A: sums and ranges
  float ss = 0.f, sg = 0.f, sf = 0.f, mns = INFINITY, mxs = -INFINITY, mng = INFINITY, mxg = -INFINITY;
  for (int i = tid; i < L; i += MT) {
    const float s = S(i), g = g_[i];
    ss += s; sg += g;
    mns = fminf(mns, s); mxs = fmaxf(mxs, s); mng = fminf(mng, g); mxg = fmaxf(mxg, g);
    if (f_) sf += f_[i];
  }
  ss = block_sum(ss, sh); sg = block_sum(sg, sh); sf = block_sum(sf, sh);
  mxs = block_max(mxs, sh); mxg = block_max(mxg, sh);
  mns = -block_max(-mns, sh); mng = -block_max(-mng, sh);
  const float mean_s = ss / (float)L, mean_g = sg / (float)L;
  const float rs = 1.f / (mxs - mns), rg = 1.f / (mxg - mng);

  // pass B: centred second moments, sums of the min-max normalised maps
  float qs = 0.f, qg = 0.f, ns = 0.f, ng = 0.f;
  for (int i = tid; i < L; i += MT) {
    const float s = S(i), g = g_[i];
    const float ds = s - mean_s, dg = g - mean_g;
    qs = fmaf(ds, ds, qs); qg = fmaf(dg, dg, qg);
    ns += (s - mns) * rs; ng += (g - mng) * rg;
  }
  qs = block_sum(qs, sh); qg = block_sum(qg, sh); ns = block_sum(ns, sh); ng = block_sum(ng, sh);
  const float std_s = sqrtf(qs / (float)(L - 1)), std_g = sqrtf(qg / (float)(L - 1));   // torch.std: unbiased

  // pass C: the four metrics' sums
  float kl = 0.f, ab = 0.f, aa = 0.f, bb = 0.f, sim = 0.f, ns_f = 0.f;
  for (int i = tid; i < L; i += MT) {
    const float s = S(i), g = g_[i];
    const float sp = s / ss, gp = g / sg;
    kl += gp * logf(eps + gp / (sp + eps));
    const float sz = (s - mean_s) / std_s, gz = (g - mean_g) / std_g;
    ab = fmaf(sz, gz, ab); aa = fmaf(sz, sz, aa); bb = fmaf(gz, gz, bb);
    sim += fminf((s - mns) * rs / ns, (g - mng) * rg / ng);
    if (f_) ns_f += (s - mean_s) / (std_s + eps) * f_[i];
  }
  kl = block_sum(kl, sh); ab = block_sum(ab, sh); aa = block_sum(aa, sh); bb = block_sum(bb, sh);
  sim = block_sum(sim, sh); ns_f = block_sum(ns_f, sh);
  if (tid == 0) {
    out[n * 4 + 0] = kl;
    out[n * 4 + 1] = ab / sqrtf(aa * bb);
    out[n * 4 + 2] = sim;
    out[n * 4 + 3] = f_ ? ns_f / sf : 0.f;
  }
}

}  // namespace mspi

extern "C" int mspi_saliency_metrics(const float* pred, const float* gt, const float* fix, float* out, int32_t N, int32_t L,
                                     int32_t pred_is_log, mspi_stream_t stream) {
  MSPI_REQUIRE(pred && gt && out && N > 0 && L > 1, "mspi_saliency_metrics: bad argument");
  hipLaunchKernelGGL(mspi::saliency_metrics_kernel, dim3(N), dim3(mspi::MT), 0, (hipStream_t)stream, pred, gt, fix, out, L,
                     pred_is_log);
  return mspi::check_launch("mspi_saliency_metrics");
}
