// Clip-loop pre-processing on the GPU (inference.py:24-63 and :154-165 upstream, where they
// are torchaudio / torchvision-on-PIL calls made once per sliding window on the host).
//
//  * log-spectrogram windows: the 16 kHz mono wave of a video sits in HBM once; every window of a batch is one grid row.
//    Spectrogram(n_fft=512, hop=160): Hann window, centre + reflect padding, power 2 -> log(p + 1e-6) -> per time column
//    standardisation over the 257 bins (unbiased std) -> crop / pad with 0.02 to Wa columns.  One workgroup per
//    (window, column); the 512-point DFT is evaluated directly against an LDS twiddle table with fp64 accumulation
//    (263 k FMA per column -- noise next to one conv), so there is no FFT plan and no intermediate in memory.
//  * frame resize + normalise: PIL's 8-bit bilinear resampling (antialiased: support scales with the shrink factor),
//    horizontal pass then vertical pass in its 22-bit fixed point with a uint8 intermediate -- integer arithmetic, so
//    the result is PIL's bit for bit -- followed by /255, -mean, /std into the NCHW fp32 frame the model takes.
#include "common.h"

namespace mspi {

constexpr int NFFT = 512, NBIN = 257;

__global__ __launch_bounds__(256) void logspec_kernel(const float* __restrict__ wave, long n_wave,
                                                      const int* __restrict__ seg,   // [B][3] = start, length, reversed
                                                      const float* __restrict__ win, float* __restrict__ out, int Wa, int hop,
                                                      float pad_value) {
  __shared__ double2 tw[NFFT];
  __shared__ float xs[NFFT];
  __shared__ float red[8];
  const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int start = seg[b * 3], len = seg[b * 3 + 1], rev = seg[b * 3 + 2];
  const int nframes = len > 0 ? 1 + len / hop : 0;
  float* o = out + (long)b * NBIN * Wa + f;
  if (f >= nframes) {        // uniform per block
    o[(long)tid * Wa] = pad_value;
    if (tid == 0) o[256L * Wa] = pad_value;
    return;
  }
  for (int j = tid; j < NFFT; j += 256) {
    double s, c;
    sincospi(2.0 * j / NFFT, &s, &c);
    tw[j] = make_double2(c, -s);
    int i = f * hop - NFFT / 2 + j;           // centre=True: the frame is centred on sample f*hop of the segment
    if (i < 0) i = -i;                        // reflect padding (requires len > NFFT/2, checked on the host)
    if (i >= len) i = 2 * (len - 1) - i;
    const long src = rev ? (long)start + len - 1 - i : (long)start + i;
    xs[j] = wave[src] * win[j];
  }
  __syncthreads();
  float lp[2];
  const int nb = tid == 0 ? 2 : 1;
  for (int q = 0; q < nb; ++q) {
    const int k = q == 0 ? tid : 256;
    double re = 0.0, im = 0.0;
#pragma unroll 8
    for (int n = 0; n < NFFT; ++n) {
      const double2 t = tw[(k * n) & (NFFT - 1)];
      const double x = (double)xs[n];
      re = fma(x, t.x, re);
      im = fma(x, t.y, im);
    }
    const float p = (float)(re * re + im * im);
    lp[q] = logf(p + 1e-6f);
  }
  // mean / unbiased std over the 257 bins of this column
  float s = lp[0] + (tid == 0 ? lp[1] : 0.f);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)NBIN;
  float d0 = lp[0] - mean, d1 = tid == 0 ? lp[1] - mean : 0.f;
  float v = wave_sum(d0 * d0 + d1 * d1);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = v;
  __syncthreads();
  const float sd = sqrtf((red[4] + red[5] + red[6] + red[7]) / (float)(NBIN - 1));
  const float inv = 1.f / (sd + 1e-6f);
  o[(long)tid * Wa] = d0 * inv;
  if (tid == 0) o[256L * Wa] = d1 * inv;
}

// PIL ImagingResample, 8 bits per channel: out = clip8((2^21 + sum_i in[xmin+i] * kk[i]) >> 22).
// Horizontal pass: src u8 [H][Win][3] (interleaved RGB) -> tmp u8 [H][Wout][3].
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                         int H, int Win, int Wout) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)H * Wout) return;
  const int x = (int)(idx % Wout), y = (int)(idx / Wout);
  const int xmin = bounds[2 * x], n = bounds[2 * x + 1];
  const int* k = kk + (long)x * ksize;
  const unsigned char* row = src + ((long)y * Win + xmin) * 3;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int i = 0; i < n; ++i) {
    const int w = k[i];
    s0 += row[3 * i] * w; s1 += row[3 * i + 1] * w; s2 += row[3 * i + 2] * w;
  }
  unsigned char* o = dst + idx * 3;
  o[0] = (unsigned char)min(255, max(0, s0 >> 22));
  o[1] = (unsigned char)min(255, max(0, s1 >> 22));
  o[2] = (unsigned char)min(255, max(0, s2 >> 22));
}

// Vertical pass + ToTensor + Normalize: tmp u8 [Hin][W][3] -> out fp32 [3][Hout][W] (plane stride out_sC floats).
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const unsigned char* __restrict__ src, float* __restrict__ out,
                                                              long out_sC, const int* __restrict__ bounds,
                                                              const int* __restrict__ kk, int ksize, int Hin, int Hout, int W,
                                                              float m0, float m1, float m2, float d0, float d1, float d2) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)Hout * W) return;
  const int x = (int)(idx % W), y = (int)(idx / W);
  const int ymin = bounds[2 * y], n = bounds[2 * y + 1];
  const int* k = kk + (long)y * ksize;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int i = 0; i < n; ++i) {
    const unsigned char* p = src + ((long)(ymin + i) * W + x) * 3;
    const int w = k[i];
    s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
  }
  const float v0 = (float)min(255, max(0, s0 >> 22)), v1 = (float)min(255, max(0, s1 >> 22)), v2 = (float)min(255, max(0, s2 >> 22));
  out[idx] = (v0 / 255.f - m0) / d0;                 // ToTensor then Normalize, in that order and with true divisions
  out[out_sC + idx] = (v1 / 255.f - m1) / d1;
  out[2 * out_sC + idx] = (v2 / 255.f - m2) / d2;
}

}  // namespace mspi

using namespace mspi;

extern "C" int mspi_logspec_fwd(const float* wave, int64_t n_wave, const int32_t* seg, const int32_t* seg_host, int32_t B,
                                const float* window, float* out, int32_t Wa, mspi_stream_t stream) {
  MSPI_REQUIRE(wave && seg && seg_host && window && out && B > 0 && B < 65536 && Wa > 0 && n_wave > 0,
               "mspi_logspec_fwd: bad argument");
  for (int b = 0; b < B; ++b) {     // the host copy of the segment table is what makes the bounds checkable before launch
    const long st = seg_host[3 * b], len = seg_host[3 * b + 1];
    MSPI_REQUIRE(st >= 0 && len >= 0 && st + len <= n_wave, "mspi_logspec_fwd: window %d = [%ld, %ld) outside the %ld-sample wave",
                 b, st, st + len, (long)n_wave);
    MSPI_REQUIRE(len == 0 || len > NFFT / 2, "mspi_logspec_fwd: window %d has %ld samples; reflect padding needs more than %d",
                 b, len, NFFT / 2);
  }
  hipLaunchKernelGGL(logspec_kernel, dim3(Wa, B), dim3(256), 0, (hipStream_t)stream, wave, (long)n_wave, seg, window, out, Wa,
                     160, 0.02f);
  return check_launch("mspi_logspec_fwd");
}

extern "C" int mspi_resize_norm_fwd(const unsigned char* rgb, int32_t Hin, int32_t Win, unsigned char* tmp, float* out,
                                    int64_t out_plane_stride, int32_t Hout, int32_t Wout, const int32_t* hb,
                                    const int32_t* hk, int32_t hks, const int32_t* vb, const int32_t* vk, int32_t vks,
                                    const float* mean3_host, const float* std3_host, mspi_stream_t stream) {
  MSPI_REQUIRE(rgb && tmp && out && hb && hk && vb && vk && mean3_host && std3_host, "mspi_resize_norm_fwd: null argument");
  MSPI_REQUIRE(Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && hks > 0 && vks > 0 && out_plane_stride >= (int64_t)Hout * Wout,
               "mspi_resize_norm_fwd: bad extent");
  hipStream_t s = (hipStream_t)stream;
  const long n1 = (long)Hin * Wout, n2 = (long)Hout * Wout;
  hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, rgb, tmp, hb, hk, hks, Hin, Win, Wout);
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, tmp, out, (long)out_plane_stride,
                     vb, vk, vks, Hin, Hout, Wout, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1],
                     std3_host[2]);
  return check_launch("mspi_resize_norm_fwd");
}
