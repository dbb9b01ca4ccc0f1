// Saliency-map post-processing on the device (reference: inference.py:66-69,85-89, OpenCV on the host):
//   GaussianBlur 11x11 (sigma from ksize = 2.0, BORDER_REFLECT_101) -> exp -> bilinear resize to (Ho,Wo)
//   (pixel-centre aligned, edge clamped) -> min-max normalise -> round(x*255) -> uint8.
// Four tiny kernels per batch: blur+exp, resize + per-workgroup min/max, min/max of those, quantise.  One D2H copy of
// Ho*Wo bytes per map replaces the reference's fp32 map download + five OpenCV passes.
// The per-map min/max is a two-stage reduction through the workspace, NOT atomics: 1,200 workgroups per map hitting two
// addresses serialise at the memory side (measured: +1.4 ms per batch of 8 maps, a tenth of the whole forward).
#include "common.h"

namespace mspi {

__constant__ float c_gauss11[11];

__global__ __launch_bounds__(256) void blur_exp_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W) {
  const int n = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * W) return;
  const int h = idx / W, w = idx - h * W;
  const float* xb = x + (long)n * H * W;
  float acc = 0.f;
  for (int i = -5; i <= 5; ++i) {
    int hh = h + i;
    hh = hh < 0 ? -hh : (hh >= H ? 2 * H - 2 - hh : hh);   // reflect-101
    hh = hh < 0 ? 0 : (hh >= H ? H - 1 : hh);
    float row = 0.f;
    for (int j = -5; j <= 5; ++j) {
      int ww = w + j;
      ww = ww < 0 ? -ww : (ww >= W ? 2 * W - 2 - ww : ww);
      ww = ww < 0 ? 0 : (ww >= W ? W - 1 : ww);
      row = fmaf(c_gauss11[j + 5], xb[hh * W + ww], row);
    }
    acc = fmaf(c_gauss11[i + 5], row, acc);
  }
  y[(long)n * H * W + idx] = expf(acc);
}

// bilinear resize (cv2.INTER_LINEAR: src = (dst + 0.5) * in/out - 0.5, clamped) + this workgroup's min/max ->
// part[n][blockIdx.x][2]
__global__ __launch_bounds__(256) void resize_minmax_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                            int Ho, int Wo, float* __restrict__ part) {
  __shared__ float smin[4], smax[4];
  const int n = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  const bool ok = idx < Ho * Wo;
  if (ok) {
    const int ho = idx / Wo, wo = idx - ho * Wo;
    float fh = ((float)ho + 0.5f) * ((float)H / (float)Ho) - 0.5f;
    float fw = ((float)wo + 0.5f) * ((float)W / (float)Wo) - 0.5f;
    int h0 = (int)floorf(fh), w0 = (int)floorf(fw);
    float lh = fh - (float)h0, lw = fw - (float)w0;
    if (h0 < 0) { h0 = 0; lh = 0.f; }
    if (w0 < 0) { w0 = 0; lw = 0.f; }
    if (h0 >= H - 1) { h0 = H - 1; lh = 0.f; }
    if (w0 >= W - 1) { w0 = W - 1; lw = 0.f; }
    const int h1 = h0 < H - 1 ? h0 + 1 : h0, w1 = w0 < W - 1 ? w0 + 1 : w0;
    const float* xb = x + (long)n * H * W;
    v = (1.f - lh) * ((1.f - lw) * xb[h0 * W + w0] + lw * xb[h0 * W + w1]) +
        lh * ((1.f - lw) * xb[h1 * W + w0] + lw * xb[h1 * W + w1]);
    y[(long)n * Ho * Wo + idx] = v;
  }
  float lo = ok ? v : INFINITY, hi = ok ? v : -INFINITY;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    hi = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    float* pp = part + 2 * ((long)n * gridDim.x + blockIdx.x);
    pp[0] = lo;
    pp[1] = hi;
  }
}

// min / max over the nb workgroup results of map n (one workgroup per map; min and max are order independent)
__global__ __launch_bounds__(256) void minmax_reduce_kernel(const float* __restrict__ part, int nb, float* __restrict__ mm) {
  __shared__ float smin[4], smax[4];
  const int n = blockIdx.x;
  float lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < nb; i += 256) {
    lo = fminf(lo, part[2 * ((long)n * nb + i)]);
    hi = fmaxf(hi, part[2 * ((long)n * nb + i) + 1]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mm[2 * n] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    mm[2 * n + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  }
}

// 16 pixels per thread, one 16-byte store: `out` may be pinned HOST memory (the clip loop reads the maps there), where
// byte-sized stores cross PCIe one partial line at a time (measured: 2.3x slower clip loop than no output at all).
__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ y, const float* __restrict__ mm,
                                                       unsigned char* __restrict__ out, int L) {
  const int n = blockIdx.y;
  const int idx = (blockIdx.x * 256 + threadIdx.x) * 16;
  if (idx >= L) return;
  const float lo = mm[2 * n], hi = mm[2 * n + 1];
  const float inv = hi > lo ? hi - lo : 1.f;        // a constant map quantises to 0 everywhere (the reference's 0/0 would be NaN -> 0 too)
  const float* yp = y + (long)n * L + idx;
  unsigned char* op = out + (long)n * L + idx;
  // the vector path needs the float4 loads AND the 16-byte store aligned: the real pointers are tested, not just the offset
  if (idx + 16 <= L && ((reinterpret_cast<uintptr_t>(yp) | reinterpret_cast<uintptr_t>(op)) & 15u) == 0) {
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(yp + 4 * q);
      const unsigned b0 = (unsigned)rintf((v.x - lo) / inv * 255.f), b1 = (unsigned)rintf((v.y - lo) / inv * 255.f);   // np.round: half to even
      const unsigned b2 = (unsigned)rintf((v.z - lo) / inv * 255.f), b3 = (unsigned)rintf((v.w - lo) / inv * 255.f);
      w[q] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
    *reinterpret_cast<uint4*>(op) = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
    for (int e = 0; e < 16 && idx + e < L; ++e) op[e] = (unsigned char)rintf((yp[e] - lo) / inv * 255.f);
  }
}

}  // namespace mspi

using namespace mspi;

// workspace: blurred [N*H*W] | resized [N*Ho*Wo] | part [N*2*nb] | mm [N*2] floats, every sub-buffer on a 16-byte boundary
static size_t pp_r4(size_t floats) { return (floats + 3) / 4 * 4; }

extern "C" size_t mspi_postprocess_workspace(int32_t N, int32_t H, int32_t W, int32_t Ho, int32_t Wo) {
  const size_t nb = ((size_t)Ho * Wo + 255) / 256;
  return (pp_r4((size_t)N * H * W) + pp_r4((size_t)N * Ho * Wo) + pp_r4((size_t)N * 2 * nb) + pp_r4((size_t)N * 2)) * sizeof(float) + 64;
}

extern "C" int mspi_postprocess_u8(const float* logmap, unsigned char* out, void* workspace, int32_t N, int32_t H, int32_t W,
                                   int32_t Ho, int32_t Wo, mspi_stream_t stream) {
  MSPI_REQUIRE(logmap && out && workspace && N > 0 && H > 1 && W > 1 && Ho > 0 && Wo > 0 && N < 65536,
               "mspi_postprocess_u8: bad argument");
  static bool init = false;
  if (!init) {   // cv2.getGaussianKernel(11, sigma = 0.3*((11-1)*0.5-1)+0.8 = 2.0)
    float k[11], s = 0.f;
    for (int i = 0; i < 11; ++i) { k[i] = expf(-(float)((i - 5) * (i - 5)) / (2.f * 2.0f * 2.0f)); s += k[i]; }
    for (int i = 0; i < 11; ++i) k[i] /= s;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_gauss11), k, sizeof(k)) != hipSuccess) {
      set_error("mspi_postprocess_u8: cannot upload the Gaussian kernel");
      (void)hipGetLastError();
      return MSPI_ELAUNCH;
    }
    init = true;
  }
  hipStream_t s = (hipStream_t)stream;
  MSPI_REQUIRE(aligned16(workspace), "mspi_postprocess_u8: workspace must be 16-byte aligned");
  float* blurred = reinterpret_cast<float*>(workspace);
  float* resized = blurred + pp_r4((size_t)N * H * W);
  const int nb = (Ho * Wo + 255) / 256;
  float* part = resized + pp_r4((size_t)N * Ho * Wo);
  float* mm = part + pp_r4((size_t)N * 2 * nb);
  hipLaunchKernelGGL(blur_exp_kernel, dim3((H * W + 255) / 256, N), dim3(256), 0, s, logmap, blurred, H, W);
  hipLaunchKernelGGL(resize_minmax_kernel, dim3(nb, N), dim3(256), 0, s, blurred, resized, H, W, Ho, Wo, part);
  hipLaunchKernelGGL(minmax_reduce_kernel, dim3(N), dim3(256), 0, s, part, nb, mm);
  hipLaunchKernelGGL(quantize_kernel, dim3((Ho * Wo + 4095) / 4096, N), dim3(256), 0, s, resized, mm, out, Ho * Wo);
  return check_launch("mspi_postprocess_u8");
}
