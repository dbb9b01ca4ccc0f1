// Depthwise convolution and max pooling, channels-last, HBM-bound kernels.
//
// Lanes run over (channel-vec4 fastest, then positions): a wavefront's loads are 16 B per lane
// over consecutive channels = whole 128-B lines.  Folded BN bias + activation are applied in the
// epilogue.  Two kernels: a strip kernel for the (k,3,3)/(k,7,7) shapes that carry the traffic
// (input vectors reused along W in registers, squeeze-excite partial sums without atomics) and a
// generic one (any kernel/stride; also max-pool).  Both remap blocks so that neighbouring outputs
// share an XCD's L2.
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

// XCD-aware bijective block remap (same as the GEMM): consecutive logical blocks -- neighbouring output
// positions, which share most of their input taps -- run on ONE XCD, so the k^3 re-reads hit that XCD's L2
// instead of being fetched once per XCD (measured 9x read amplification without it).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <bool IS_MAX>
__global__ __launch_bounds__(256) void dw_kernel(const DwArgs p) {
  const int n = blockIdx.y;
  const long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
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
    if (!IS_MAX) {
      acc.x = act_apply(acc.x, p.act);
      acc.y = act_apply(acc.y, p.act);
      acc.z = act_apply(acc.z, p.act);
      acc.w = act_apply(acc.w, p.act);
    }
    const long orow = (((long)n * p.To + to) * p.Ho + ho) * p.Wo + wo;
    *reinterpret_cast<float4*>(p.y + orow * p.ldy + cv * 4) = acc;
  }
}

// Strip kernel: one thread = SW consecutive outputs along W for one channel-vec4.  Per (dt,dh) input row the
// (SW-1)*SWS+KW input vectors are loaded once and reused across the strip, the KW weight vectors once.
// POOL: per-(n,c) sums of the pre-activation output are reduced deterministically: per-thread strip sum ->
// LDS stage -> fixed-order in-block sum -> one partial row per block in pool[n][block][C] (no atomics).
template <int KW, int SWS, int SW, bool POOL>
__global__ __launch_bounds__(256) void dw_strip_kernel(const DwArgs p) {
  __shared__ float4 stage[POOL ? 256 : 1];
  const int n = blockIdx.y;
  const int nblk = gridDim.x;
  const int lb = xcd_remap(blockIdx.x, nblk);
  const long idx0 = (long)lb * 256;
  const long idx = idx0 + threadIdx.x;
  const int S = (p.Wo + SW - 1) / SW;
  const long per_sample = (long)p.To * p.Ho * S * p.CV;
  float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < per_sample) {
    const int cv = (int)(idx % p.CV);
    long pos = idx / p.CV;
    const int ws = (int)(pos % S);
    pos /= S;
    const int ho = (int)(pos % p.Ho);
    const int to = (int)(pos / p.Ho);
    const int wo0 = ws * SW;
    const int t0 = to * p.strT - p.padT, h0 = ho * p.strH - p.padH, w0 = wo0 * SWS - p.padW;
    const float* xb = p.x + ((long)n * p.T * p.H * p.W) * p.ldx + cv * 4;
    const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + cv * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[SW];
#pragma unroll
    for (int o = 0; o < SW; ++o) acc[o] = bv;
    constexpr int NIN = (SW - 1) * SWS + KW;
    for (int dt = 0; dt < p.kT; ++dt) {
      const int t = t0 + dt;
      if ((unsigned)t >= (unsigned)p.T) continue;
      for (int dh = 0; dh < p.kH; ++dh) {
        const int h = h0 + dh;
        if ((unsigned)h >= (unsigned)p.H) continue;
        const float* xr = xb + ((long)(t * p.H + h) * p.W) * p.ldx;
        const float* wr = p.w + (long)((dt * p.kH + dh) * KW) * p.C + cv * 4;
        float4 wt[KW];
#pragma unroll
        for (int k = 0; k < KW; ++k) wt[k] = *reinterpret_cast<const float4*>(wr + (long)k * p.C);
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
          const int w = w0 + j;
          float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
          if ((unsigned)w < (unsigned)p.W) xv = *reinterpret_cast<const float4*>(xr + (long)w * p.ldx);
#pragma unroll
          for (int o = 0; o < SW; ++o) {
            const int k = j - o * SWS;
            if (k >= 0 && k < KW) {
              acc[o].x = fmaf(xv.x, wt[k].x, acc[o].x);
              acc[o].y = fmaf(xv.y, wt[k].y, acc[o].y);
              acc[o].z = fmaf(xv.z, wt[k].z, acc[o].z);
              acc[o].w = fmaf(xv.w, wt[k].w, acc[o].w);
            }
          }
        }
      }
    }
    const long orow = (((long)n * p.To + to) * p.Ho + ho) * p.Wo + wo0;
#pragma unroll
    for (int o = 0; o < SW; ++o) {
      if (wo0 + o < p.Wo) {
        float4 v = acc[o];
        if (POOL) { psum.x += v.x; psum.y += v.y; psum.z += v.z; psum.w += v.w; }
        v.x = act_apply(v.x, p.act); v.y = act_apply(v.y, p.act); v.z = act_apply(v.z, p.act); v.w = act_apply(v.w, p.act);
        *reinterpret_cast<float4*>(p.y + (orow + o) * p.ldy + cv * 4) = v;
      }
    }
  }
  if (POOL) {
    stage[threadIdx.x] = psum;
    __syncthreads();
    // threads j = r, r+CV, r+2CV, ... of this block hold the same channel-vec cv_r = (idx0 + r) % CV
    for (int r = threadIdx.x; r < p.CV; r += 256) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = r; j < 256; j += p.CV) { s.x += stage[j].x; s.y += stage[j].y; s.z += stage[j].z; s.w += stage[j].w; }
      const int cvr = (int)((idx0 + r) % p.CV);
      *reinterpret_cast<float4*>(p.pool + ((long)n * nblk + lb) * p.C + cvr * 4) = s;
    }
  }
}

constexpr int DW_SW = 4;  // strip length along W

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

// which strip instantiation serves this descriptor: 0 = (kW 3, stride 1), 1 = (3, 2), 2 = (7, 1), -1 = generic kernel
static int strip_variant(const MspiDwConvDesc* d) {
  if (d->kW == 3 && d->strW == 1) return 0;
  if (d->kW == 3 && d->strW == 2) return 1;
  if (d->kW == 7 && d->strW == 1) return 2;
  return -1;
}

extern "C" int mspi_dwconv_fwd(const MspiDwConvDesc* d, const float* x, const float* w, const float* bias, float* y,
                               float* pool, mspi_stream_t stream) {
  MSPI_REQUIRE(d && x && w && y, "mspi_dwconv_fwd: null argument");
  DwArgs a;
  int rc = fill_args(d, a, "mspi_dwconv_fwd");
  if (rc) return rc;
  MSPI_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && (!bias || aligned16(bias)),
               "mspi_dwconv_fwd: pointers must be 16-B aligned");
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.pool = pool;
  hipStream_t s = (hipStream_t)stream;
  const int strip = strip_variant(d);
  MSPI_REQUIRE(!pool || strip >= 0, "mspi_dwconv_fwd: SE pooling needs a (k,3,3)/(k,7,7) kernel with W-stride 1 or 2");
  if (strip >= 0) {
    const long S = (a.Wo + DW_SW - 1) / DW_SW;
    const long per = (long)a.To * a.Ho * S * a.CV;
    dim3 grid((unsigned)((per + 255) / 256), (unsigned)a.N);
#define MSPI_DW_LAUNCH(KW, SWS)                                                                                  \
    do {                                                                                                           \
      if (pool) hipLaunchKernelGGL((dw_strip_kernel<KW, SWS, DW_SW, true>), grid, dim3(256), 0, s, a);             \
      else hipLaunchKernelGGL((dw_strip_kernel<KW, SWS, DW_SW, false>), grid, dim3(256), 0, s, a);                 \
    } while (0)
    switch (strip) {
      case 0: MSPI_DW_LAUNCH(3, 1); break;
      case 1: MSPI_DW_LAUNCH(3, 2); break;
      default: MSPI_DW_LAUNCH(7, 1); break;
    }
#undef MSPI_DW_LAUNCH
  } else {
    dim3 grid((unsigned)((a.per_sample + 255) / 256), (unsigned)a.N);
    hipLaunchKernelGGL((dw_kernel<false>), grid, dim3(256), 0, s, a);
  }
  return check_launch("mspi_dwconv_fwd");
}

extern "C" int mspi_dwconv_pool_rows(const MspiDwConvDesc* d) {
  if (!d || strip_variant(d) < 0) return -1;
  const long Wo = (d->W + 2 * d->padW - d->kW) / d->strW + 1;
  const long S = (Wo + DW_SW - 1) / DW_SW;
  const long per = (long)d->To * d->Ho * S * (d->C / 4);
  return (int)((per + 255) / 256);
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
  hipLaunchKernelGGL((dw_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("mspi_maxpool_fwd");
}
