// Depthwise convolution and max pooling, channels-last, HBM-bound kernels.
//
// Lanes run over (channel-vec4 fastest, then wo): a wavefront's loads are 16 B per lane
// over consecutive channels = whole 128-B lines, and the k^3 re-reads of a position are
// served by L1/L2 (neighbouring outputs share taps).  Folded BN bias + activation are
// applied in the epilogue; for X3D squeeze-excite blocks the per-(n,c) sums of the BN
// output are reduced in LDS and added to `pool` with one float atomic per block/channel.
#include "common.h"

namespace mspi {

struct DwArgs {
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  float* pool;
  int N, T, H, W, C;
  long ldx, ldy;
  int kT, kH, kW, strT, strH, strW, padT, padH, padW;
  int To, Ho, Wo;
  int act;
  int CV;          // C / 4
  long per_sample; // To*Ho*Wo*CV
};

template <bool POOL, bool IS_MAX>
__global__ __launch_bounds__(256) void dw_kernel(const DwArgs p) {
  extern __shared__ float lsum[];  // [C] when POOL
  const int n = blockIdx.y;
  if (POOL) {
    for (int c = threadIdx.x; c < p.C; c += 256) lsum[c] = 0.f;
    __syncthreads();
  }
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx < p.per_sample) {
    const int cv = (int)(idx % p.CV);
    long pos = idx / p.CV;
    const int wo = (int)(pos % p.Wo);
    pos /= p.Wo;
    const int ho = (int)(pos % p.Ho);
    const int to = (int)(pos / p.Ho);
    const int t0 = to * p.strT - p.padT, h0 = ho * p.strH - p.padH, w0 = wo * p.strW - p.padW;
    const float* xb = p.x + ((long)n * p.T * p.H * p.W) * p.ldx + cv * 4;
    float4 acc;
    if (IS_MAX) {
      acc = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    } else {
      acc = p.bias ? *reinterpret_cast<const float4*>(p.bias + cv * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int dt = 0; dt < p.kT; ++dt) {
      const int t = t0 + dt;
      if ((unsigned)t >= (unsigned)p.T) continue;
      for (int dh = 0; dh < p.kH; ++dh) {
        const int h = h0 + dh;
        if ((unsigned)h >= (unsigned)p.H) continue;
        const float* xr = xb + ((long)(t * p.H + h) * p.W) * p.ldx;
        const float* wr = IS_MAX ? nullptr : p.w + (long)((dt * p.kH + dh) * p.kW) * p.C + cv * 4;
        for (int dw = 0; dw < p.kW; ++dw) {
          const int w = w0 + dw;
          if ((unsigned)w >= (unsigned)p.W) continue;
          const float4 xv = *reinterpret_cast<const float4*>(xr + (long)w * p.ldx);
          if (IS_MAX) {
            acc.x = fmaxf(acc.x, xv.x);
            acc.y = fmaxf(acc.y, xv.y);
            acc.z = fmaxf(acc.z, xv.z);
            acc.w = fmaxf(acc.w, xv.w);
          } else {
            const float4 wv = *reinterpret_cast<const float4*>(wr + (long)dw * p.C);
            acc.x = fmaf(xv.x, wv.x, acc.x);
            acc.y = fmaf(xv.y, wv.y, acc.y);
            acc.z = fmaf(xv.z, wv.z, acc.z);
            acc.w = fmaf(xv.w, wv.w, acc.w);
          }
        }
      }
    }
    if (POOL) {
      atomicAdd(&lsum[cv * 4 + 0], acc.x);
      atomicAdd(&lsum[cv * 4 + 1], acc.y);
      atomicAdd(&lsum[cv * 4 + 2], acc.z);
      atomicAdd(&lsum[cv * 4 + 3], acc.w);
    }
    if (!IS_MAX) {
      acc.x = act_apply(acc.x, p.act);
      acc.y = act_apply(acc.y, p.act);
      acc.z = act_apply(acc.z, p.act);
      acc.w = act_apply(acc.w, p.act);
    }
    const long orow = (((long)n * p.To + to) * p.Ho + ho) * p.Wo + wo;
    *reinterpret_cast<float4*>(p.y + orow * p.ldy + cv * 4) = acc;
  }
  if (POOL) {
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += 256) {
      const float v = lsum[c];
      if (v != 0.f) atomicAdd(p.pool + (long)n * p.C + c, v);
    }
  }
}

static int fill_args(const MspiDwConvDesc* d, DwArgs& a, const char* who) {
  MSPI_REQUIRE(d->N > 0 && d->T > 0 && d->H > 0 && d->W > 0 && d->C > 0, "%s: empty extent", who);
  MSPI_REQUIRE((d->C & 3) == 0 && (d->ldx & 3) == 0 && (d->ldy & 3) == 0 && d->ldx >= d->C && d->ldy >= d->C,
               "%s: C/ldx/ldy must be multiples of 4 with ld >= C", who);
  const int To = (d->T + 2 * d->padT - d->kT) / d->strT + 1;
  const int Ho = (d->H + 2 * d->padH - d->kH) / d->strH + 1;
  const int Wo = (d->W + 2 * d->padW - d->kW) / d->strW + 1;
  MSPI_REQUIRE(To == d->To && Ho == d->Ho && Wo == d->Wo && To > 0 && Ho > 0 && Wo > 0,
               "%s: output extent (%d,%d,%d) does not match formula (%d,%d,%d)", who, d->To, d->Ho, d->Wo, To, Ho, Wo);
  a.N = d->N; a.T = d->T; a.H = d->H; a.W = d->W; a.C = d->C;
  a.ldx = d->ldx; a.ldy = d->ldy;
  a.kT = d->kT; a.kH = d->kH; a.kW = d->kW;
  a.strT = d->strT; a.strH = d->strH; a.strW = d->strW;
  a.padT = d->padT; a.padH = d->padH; a.padW = d->padW;
  a.To = To; a.Ho = Ho; a.Wo = Wo; a.act = d->act;
  a.CV = d->C / 4;
  a.per_sample = (long)To * Ho * Wo * a.CV;
  MSPI_REQUIRE((a.per_sample + 255) / 256 < (1L << 31) && d->N < 65536, "%s: grid too large", who);
  return MSPI_OK;
}

}  // namespace mspi

using namespace mspi;

extern "C" int mspi_dwconv_fwd(const MspiDwConvDesc* d, const float* x, const float* w, const float* bias, float* y,
                               float* pool, mspi_stream_t stream) {
  MSPI_REQUIRE(d && x && w && y, "mspi_dwconv_fwd: null argument");
  DwArgs a;
  int rc = fill_args(d, a, "mspi_dwconv_fwd");
  if (rc) return rc;
  MSPI_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && (!bias || aligned16(bias)),
               "mspi_dwconv_fwd: pointers must be 16-B aligned");
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.pool = pool;
  dim3 grid((unsigned)((a.per_sample + 255) / 256), (unsigned)a.N);
  if (pool)
    hipLaunchKernelGGL((dw_kernel<true, false>), grid, dim3(256), (size_t)a.C * sizeof(float), (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((dw_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("mspi_dwconv_fwd");
}

extern "C" int mspi_maxpool_fwd(const MspiDwConvDesc* d, const float* x, float* y, mspi_stream_t stream) {
  MSPI_REQUIRE(d && x && y, "mspi_maxpool_fwd: null argument");
  DwArgs a;
  int rc = fill_args(d, a, "mspi_maxpool_fwd");
  if (rc) return rc;
  MSPI_REQUIRE(aligned16(x) && aligned16(y), "mspi_maxpool_fwd: pointers must be 16-B aligned");
  // PyTorch requires pad <= kernel/2, so every window holds at least one valid element
  MSPI_REQUIRE(2 * d->padT <= d->kT && 2 * d->padH <= d->kH && 2 * d->padW <= d->kW, "mspi_maxpool_fwd: pad > kernel/2");
  a.x = x; a.w = nullptr; a.bias = nullptr; a.y = y; a.pool = nullptr;
  dim3 grid((unsigned)((a.per_sample + 255) / 256), (unsigned)a.N);
  hipLaunchKernelGGL((dw_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("mspi_maxpool_fwd");
}
