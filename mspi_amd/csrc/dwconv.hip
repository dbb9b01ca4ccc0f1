// Depthwise convolution and max pooling, channels-last, HBM-bound kernels.
//
// Lanes run over (channel-vec4 fastest, then positions): a wavefront's loads are 16 B per lane
// over consecutive channels = whole 128-B lines.  Folded BN bias + activation are applied in the
// epilogue.  Two kernels: a strip kernel for the (k,3,3)/(k,7,7) shapes that carry the traffic
// (input vectors reused along W in registers, squeeze-excite partial sums without atomics) and a
// generic one (any kernel/stride; also max-pool).  Both remap blocks so that neighbouring outputs
// share an XCD's L2.
#include "common.h"
#include <stdlib.h>

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
  int unit_xcd;    // one unit (sample / sample x channel group) per XCD (A/B switch MSPI_DW_UNIT_XCD=0)
};

// XCD-aware bijective block remap (same as the GEMM): consecutive logical blocks -- neighbouring output
// positions, which share most of their input taps -- run on ONE XCD, so the k^3 re-reads hit that XCD's L2
// instead of being fetched once per XCD (measured 9x read amplification without it).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// One work UNIT (a sample, or a sample x channel group) per XCD: workgroups are dealt to the 8 XCDs round-robin in linear
// launch order, so workgroup L of a group of eight units takes unit L % 8 and block L / 8 of it.  An XCD then streams its
// unit front to back: the window re-reads stay in ITS L2 and HBM sees every record once -- against 1.6-2.3x the algorithmic
// bytes (PMC) when every unit is cut into eight position slices that each need their neighbours' halo.  Needs the unit
// count to be a multiple of 8; otherwise the per-unit slice remap above.
__device__ __forceinline__ void unit_per_xcd(long L, int nblk, int units, int& unit, int& lb, int enabled) {
  if (enabled && (units & 7) == 0) {
    const long q = L / (8L * nblk), r = L - q * 8L * nblk;
    unit = (int)(q * 8 + (r & 7));
    lb = (int)(r >> 3);
  } else {
    unit = (int)(L / nblk);
    lb = xcd_remap((int)(L - (long)unit * nblk), nblk);
  }
}

template <bool IS_MAX>
__global__ __launch_bounds__(256) void dw_kernel(const DwArgs p) {
  int n, lb0;
  unit_per_xcd((long)blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, n, lb0, p.unit_xcd);
  // 32-bit index arithmetic (per_sample < 2^31 - 256, checked on the host): as `long` each of these divisions is a ~100-instruction
  // 64-bit sequence, four per thread -- beside a few hundred FMAs
  const unsigned idx = (unsigned)lb0 * 256u + threadIdx.x;
  if (idx < (unsigned)p.per_sample) {
    const int cv = (int)(idx % (unsigned)p.CV);
    unsigned pos = idx / (unsigned)p.CV;
    const int wo = (int)(pos % (unsigned)p.Wo);
    pos /= (unsigned)p.Wo;
    const int ho = (int)(pos % (unsigned)p.Ho);
    const int to = (int)(pos / (unsigned)p.Ho);
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
      auto fin = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
        acc.x = act_apply(acc.x, ACT); acc.y = act_apply(acc.y, ACT); acc.z = act_apply(acc.z, ACT); acc.w = act_apply(acc.w, ACT);
      };
      MSPI_DISPATCH_ACT(p.act, fin)
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
  const int nblk = gridDim.x;
  int n, lb;
  unit_per_xcd((long)blockIdx.y * nblk + blockIdx.x, nblk, gridDim.y, n, lb, p.unit_xcd);
  const unsigned idx0 = (unsigned)lb * 256u;      // 32-bit index arithmetic (see dw_kernel)
  const unsigned idx = idx0 + threadIdx.x;
  const int S = (p.Wo + SW - 1) / SW;
  const unsigned per_sample = (unsigned)p.To * p.Ho * S * p.CV;
  float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < per_sample) {
    const int cv = (int)(idx % (unsigned)p.CV);
    unsigned pos = idx / (unsigned)p.CV;
    const int ws = (int)(pos % (unsigned)S);
    pos /= (unsigned)S;
    const int ho = (int)(pos % (unsigned)p.Ho);
    const int to = (int)(pos / (unsigned)p.Ho);
    const int wo0 = ws * SW;
    const int t0 = to * p.strT - p.padT, h0 = ho * p.strH - p.padH, w0 = wo0 * SWS - p.padW;
    const float* xb = p.x + ((long)n * p.T * p.H * p.W) * p.ldx + cv * 4;
    const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + cv * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[SW];
#pragma unroll
    for (int o = 0; o < SW; ++o) acc[o] = bv;
    constexpr int NIN = (SW - 1) * SWS + KW;
    // kH == KW (square kernels only reach this kernel).  The KW input rows of one dt are fetched as ONE batch of
    // unconditional loads (clamped address + select): on the small X3D maps the whole grid is a single round of
    // blocks, so the kernel's length is (number of dependent load phases) x (loaded latency) -- 3 phases, not 9.
    for (int dt = 0; dt < p.kT; ++dt) {
      const int t = t0 + dt;
      if ((unsigned)t >= (unsigned)p.T) continue;
      constexpr int RB = KW == 3 ? 3 : 1;   // rows per load batch (7x7: 1 and a ROLLED loop -- unrolled, the compiler hoists
                                            // all 70 input vectors, spills, and the 14x14 maps run 4x slower)
#pragma unroll 1
      for (int dh0 = 0; dh0 < KW; dh0 += RB) {
        float4 xin[RB][NIN];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          if (dh0 + b >= KW) continue;
          const int h = h0 + dh0 + b;
          const bool hv = (unsigned)h < (unsigned)p.H;
          const float* xr = xb + ((long)(t * p.H + (hv ? h : 0)) * p.W) * p.ldx;
#pragma unroll
          for (int j = 0; j < NIN; ++j) {
            const int w = w0 + j;
            const bool ok = hv && (unsigned)w < (unsigned)p.W;
            const float4 v = *reinterpret_cast<const float4*>(xr + (long)(ok ? w : 0) * p.ldx);
            xin[b][j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          if (dh0 + b >= KW) continue;
          const float* wr = p.w + (long)((dt * KW + dh0 + b) * KW) * p.C + cv * 4;
#pragma unroll
          for (int k = 0; k < KW; ++k) {
            const float4 wv = *reinterpret_cast<const float4*>(wr + (long)k * p.C);
#pragma unroll
            for (int o = 0; o < SW; ++o) {
              const float4 xv = xin[b][o * SWS + k];
              acc[o].x = fmaf(xv.x, wv.x, acc[o].x);
              acc[o].y = fmaf(xv.y, wv.y, acc[o].y);
              acc[o].z = fmaf(xv.z, wv.z, acc[o].z);
              acc[o].w = fmaf(xv.w, wv.w, acc[o].w);
            }
          }
        }
      }
    }
    const long orow = (((long)n * p.To + to) * p.Ho + ho) * p.Wo + wo0;
    auto fin = [&](auto act_c) {      // the activation as a compile-time constant inside the loop (common.h)
      constexpr int ACT = decltype(act_c)::value;
#pragma unroll
      for (int o = 0; o < SW; ++o) {
        if (wo0 + o < p.Wo) {
          float4 v = acc[o];
          if (POOL) { psum.x += v.x; psum.y += v.y; psum.z += v.z; psum.w += v.w; }
          v.x = act_apply(v.x, ACT); v.y = act_apply(v.y, ACT); v.z = act_apply(v.z, ACT); v.w = act_apply(v.w, ACT);
          *reinterpret_cast<float4*>(p.y + (orow + o) * p.ldy + cv * 4) = v;
        }
      }
    };
    MSPI_DISPATCH_ACT(p.act, fin)
  }
  if (POOL) {
    stage[threadIdx.x] = psum;
    __syncthreads();
    // threads j = r, r+CV, r+2CV, ... of this block hold the same channel-vec cv_r = (idx0 + r) % CV
    for (int r = threadIdx.x; r < p.CV; r += 256) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = r; j < 256; j += p.CV) { s.x += stage[j].x; s.y += stage[j].y; s.z += stage[j].z; s.w += stage[j].w; }
      const int cvr = (int)((idx0 + r) % (unsigned)p.CV);
      *reinterpret_cast<float4*>(p.pool + ((long)n * nblk + lb) * p.C + cvr * 4) = s;
    }
  }
}

// Tile kernel: one thread = SH x SW outputs (rows x columns) of one channel-vec4.  Why: the strip kernel issues
// (NIN + KW) 16-B global loads per input row for SW*KW FMA-vec4 -- 17.5 loads per output for 7x7, 20 for 3x3x3 --
// and runs at the L1/TA rate (1.6-2.5 TB/s algorithmic, profiles/r01), not at HBM's.  Here
//   * an input row is loaded once and feeds up to SH output rows (7x7, SH=2, SW=7: 7.4 loads per output),
//   * the weights come from LDS (staged once per block for the block's <= 32 channel-vec4s), not from L1,
//   * loads are unconditional (clamped address + select), so the compiler batches them under one wait.
// grid = (position blocks, channel groups, N); thread -> (channel-vec4 = tid % CG fastest, position = tid / CG).
// POOL partial rows: pool[n][position block][C], each channel group fills its own channel slice.
template <int K, int STR, int SW, int SH, bool POOL>
__global__ __launch_bounds__(256) void dw_tile_kernel(const DwArgs p, int CG, int PB, int HS, int S) {
  extern __shared__ float4 dw_smem[];          // weights [taps][CG], then (POOL) stage[256]
  const int taps = p.kT * K * K;
  float4* wl = dw_smem;
  float4* stage = dw_smem + taps * CG;
  const int tid = threadIdx.x;
  const int nblk = gridDim.x;
  int unit, lb;      // unit = (sample, channel group)
  unit_per_xcd(((long)blockIdx.z * gridDim.y + blockIdx.y) * nblk + blockIdx.x, nblk, gridDim.y * gridDim.z, unit, lb, p.unit_xcd);
  const int n = unit / gridDim.y, g = unit - n * gridDim.y;
  for (int i = tid; i < taps * CG; i += 256) {
    const int tap = i / CG, c = i - tap * CG;
    wl[i] = *reinterpret_cast<const float4*>(p.w + (long)tap * p.C + (g * CG + c) * 4);
  }
  __syncthreads();
  const int cvl = tid % CG, pl = tid / CG;
  const unsigned pos = (unsigned)lb * PB + pl;      // 32-bit index arithmetic (see dw_kernel)
  const long npos = (long)p.To * HS * S;
  float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pl < PB && pos < npos) {
    const int ws = (int)(pos % (unsigned)S);
    const int hs = (int)((pos / (unsigned)S) % (unsigned)HS);
    const int to = (int)(pos / ((unsigned)S * HS));
    const int cv = g * CG + cvl;
    const int wo0 = ws * SW, ho0 = hs * SH;
    const int t0 = to * p.strT - p.padT, h0 = ho0 * STR - p.padH, w0 = wo0 * STR - p.padW;
    const float* xb = p.x + ((long)n * p.T * p.H * p.W) * p.ldx + cv * 4;
    const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + cv * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[SH][SW];
#pragma unroll
    for (int sh = 0; sh < SH; ++sh)
#pragma unroll
      for (int o = 0; o < SW; ++o) acc[sh][o] = bv;
    constexpr int NIN = (SW - 1) * STR + K, RIN = (SH - 1) * STR + K;
    for (int dt = 0; dt < p.kT; ++dt) {
      const int t = t0 + dt;
      if ((unsigned)t >= (unsigned)p.T) continue;
      const float4* wt = wl + (dt * K * K) * CG + cvl;
#pragma unroll 1   // rolled: unrolled, the compiler hoists every row's loads and spills (1.6 KB scratch/lane, occupancy 1)
      for (int r = 0; r < RIN; ++r) {
        const int h = h0 + r;
        const bool hv = (unsigned)h < (unsigned)p.H;
        const float* xr = xb + ((long)(t * p.H + (hv ? h : 0)) * p.W) * p.ldx;
        float4 xin[NIN];
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
          const int w = w0 + j;
          const bool ok = hv && (unsigned)w < (unsigned)p.W;
          const float4 v = *reinterpret_cast<const float4*>(xr + (long)(ok ? w : 0) * p.ldx);
          xin[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int sh = 0; sh < SH; ++sh) {
          const int dh = r - sh * STR;
          if (dh < 0 || dh >= K) continue;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float4 wv = wt[(dh * K + k) * CG];
#pragma unroll
            for (int o = 0; o < SW; ++o) {
              const float4 xv = xin[o * STR + k];
              if (K == 7) {
                fma4(acc[sh][o], xv, wv);      // packed fp32 FMA (common.h): this kernel is VALU-bound at 7x7
              } else {
                acc[sh][o].x = fmaf(xv.x, wv.x, acc[sh][o].x);
                acc[sh][o].y = fmaf(xv.y, wv.y, acc[sh][o].y);
                acc[sh][o].z = fmaf(xv.z, wv.z, acc[sh][o].z);
                acc[sh][o].w = fmaf(xv.w, wv.w, acc[sh][o].w);
              }
            }
          }
        }
      }
    }
    auto fin = [&](auto act_c) {      // the activation as a compile-time constant inside the loops (common.h)
      constexpr int ACT = decltype(act_c)::value;
#pragma unroll
      for (int sh = 0; sh < SH; ++sh) {
        if (ho0 + sh >= p.Ho) continue;
        const long orow = (((long)n * p.To + to) * p.Ho + ho0 + sh) * p.Wo + wo0;
#pragma unroll
        for (int o = 0; o < SW; ++o) {
          if (wo0 + o < p.Wo) {
            float4 v = acc[sh][o];
            if (POOL) { psum.x += v.x; psum.y += v.y; psum.z += v.z; psum.w += v.w; }
            v.x = act_apply(v.x, ACT); v.y = act_apply(v.y, ACT); v.z = act_apply(v.z, ACT); v.w = act_apply(v.w, ACT);
            *reinterpret_cast<float4*>(p.y + (orow + o) * p.ldy + cv * 4) = v;
          }
        }
      }
    };
    MSPI_DISPATCH_ACT(p.act, fin)
  }
  if (POOL) {
    stage[tid] = psum;
    __syncthreads();
    if (tid < CG) {   // fixed order over the block's positions: bitwise reproducible
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int q = 0; q < PB; ++q) { const float4 v = stage[q * CG + tid]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
      *reinterpret_cast<float4*>(p.pool + ((long)n * nblk + lb) * p.C + (g * CG + tid) * 4) = s;
    }
  }
}


// LDS-staged kernel (stride 1, square K x K spatial kernels, any temporal extent): a block owns TT output frames x a
// TH x TW spatial tile x CG channel-vec4s.  Why: the strip / tile kernels above fetch every input 13-17 times through
// L1 (they run at the texture-address rate, 1.4-2.4 TB/s algorithmic) in kT dependent load phases.  Here
//   * the block's input region ((TT+kT-1) x (TH+K-1) x (TW+K-1) pixels x CG vec4, zero-filled outside the map) is brought
//     into LDS ONCE by a single batch of independent, coalesced 16-B loads (one latency phase): (1 + (kT-1)/TT) x halo
//     = 2-2.5 L1 requests per output instead of 13.5-17.5;
//   * outputs are then computed from LDS with the same register sliding window along W (SW outputs per thread and
//     row), weights from LDS as in the tile kernel, no bounds checks in the inner loops.
// grid = (tt x th x tw tiles [XCD-remapped], channel groups, N); thread -> (channel-vec4 = tid % CG, strip = tid / CG).
// POOL partial rows: pool[n][tile][C], each channel group fills its own slice, fixed summation order.
__device__ __forceinline__ int fdiv(int a, float inv_b) { return __float2int_rz(((float)a + 0.5f) * inv_b); }

template <int K, int SW, bool POOL>
__global__ __launch_bounds__(256) void dw_lds_kernel(const DwArgs p, int CG, int TT, int TH, int nsw, int nTh, int nTw,
                                                     float inv_cg, float inv_rw, float inv_rh) {
  extern __shared__ float4 dwl[];
  const int kT = p.kT;
  const int TWp = nsw * SW;
  const int RT = TT + kT - 1, RH = TH + K - 1, RW = TWp + K - 1;
  const int taps = kT * K * K;
  float4* xin = dwl;                              // [RT][RH][RW][CG]
  float4* wl = dwl + RT * RH * RW * CG;           // [taps][CG]
  float4* stage = wl + taps * CG;                 // POOL: [256]
  const int tid = threadIdx.x;
  const int n = blockIdx.z, g = blockIdx.y;
  const int nblk = gridDim.x;
  const int lb = xcd_remap(blockIdx.x, nblk);   // (unit-per-XCD measured slower here: a unit is a <= 128-B slice of every pixel)
  const int tw = lb % nTw, th = (lb / nTw) % nTh, tt = lb / (nTw * nTh);
  const int to0 = tt * TT, ho0 = th * TH, wo0 = tw * TWp;
  const int t0 = to0 - p.padT, h0 = ho0 - p.padH, w0 = wo0 - p.padW;
  for (int i = tid; i < taps * CG; i += 256) {
    const int tap = fdiv(i, inv_cg), c = i - tap * CG;
    wl[i] = *reinterpret_cast<const float4*>(p.w + (long)tap * p.C + (g * CG + c) * 4);
  }
  const float* xb = p.x + ((long)n * p.T * p.H * p.W) * p.ldx + g * CG * 4;
  const int E = RT * RH * RW * CG;
  constexpr int U = 8;
  for (int base = tid; base < E; base += 256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + u * 256;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < E) {
        const int px = fdiv(e, inv_cg), c = e - px * CG;
        const int q = fdiv(px, inv_rw), rw = px - q * RW;
        const int dt = fdiv(q, inv_rh), rh = q - dt * RH;
        const int t = t0 + dt, h = h0 + rh, w = w0 + rw;
        if ((unsigned)t < (unsigned)p.T && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W)
          v[u] = *reinterpret_cast<const float4*>(xb + (((long)t * p.H + h) * p.W + w) * p.ldx + c * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + u * 256;
      if (e < E) xin[e] = v[u];
    }
  }
  __syncthreads();
  const int tpc = 256 / CG;                       // strips in flight per pass
  const int cvl = tid % CG, sp = tid / CG;
  const int cv = g * CG + cvl;
  const int strips = TT * TH * nsw;
  float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sp < tpc) {
    const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + cv * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = sp; s < strips; s += tpc) {
      const int cs = s % nsw, r = (s / nsw) % TH, ot = s / (nsw * TH);
      const int to = to0 + ot, ho = ho0 + r, wo = wo0 + cs * SW;
      if (to >= p.To || ho >= p.Ho || wo >= p.Wo) continue;
      float4 acc[SW];
#pragma unroll
      for (int o = 0; o < SW; ++o) acc[o] = bv;
      for (int dt = 0; dt < kT; ++dt) {
        if ((unsigned)(to - p.padT + dt) >= (unsigned)p.T) continue;      // that frame of the region is all zeros
        const float4* wt = wl + (dt * K * K) * CG + cvl;
#pragma unroll
        for (int dh = 0; dh < K; ++dh) {
          const float4* row = xin + ((((ot + dt) * RH + r + dh) * RW) + cs * SW) * CG + cvl;
          float4 x[SW + K - 1];
#pragma unroll
          for (int j = 0; j < SW + K - 1; ++j) x[j] = row[j * CG];
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float4 wv = wt[(dh * K + k) * CG];
#pragma unroll
            for (int o = 0; o < SW; ++o) {
              if (K == 7) {
                fma4(acc[o], x[o + k], wv);      // packed fp32 FMA (common.h): 7x7 -7 ... -9 % on 14 / 28-wide maps; 5x5x5 +3 ... +11 % (not used)
              } else {
                acc[o].x = fmaf(x[o + k].x, wv.x, acc[o].x);
                acc[o].y = fmaf(x[o + k].y, wv.y, acc[o].y);
                acc[o].z = fmaf(x[o + k].z, wv.z, acc[o].z);
                acc[o].w = fmaf(x[o + k].w, wv.w, acc[o].w);
              }
            }
          }
        }
      }
      const long orow = (((long)n * p.To + to) * p.Ho + ho) * p.Wo + wo;
      auto fin = [&](auto act_c) {      // the activation as a compile-time constant inside the loop (common.h)
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int o = 0; o < SW; ++o) {
          if (wo + o < p.Wo) {
            float4 v = acc[o];
            if (POOL) { psum.x += v.x; psum.y += v.y; psum.z += v.z; psum.w += v.w; }
            v.x = act_apply(v.x, ACT); v.y = act_apply(v.y, ACT); v.z = act_apply(v.z, ACT); v.w = act_apply(v.w, ACT);
            *reinterpret_cast<float4*>(p.y + (orow + o) * p.ldy + cv * 4) = v;
          }
        }
      };
      MSPI_DISPATCH_ACT(p.act, fin)
    }
  }
  if (POOL) {
    stage[tid] = psum;
    __syncthreads();
    if (tid < CG) {   // fixed order over the block's strips: bitwise reproducible
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int q = 0; q < tpc; ++q) { const float4 v = stage[q * CG + tid]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      *reinterpret_cast<float4*>(p.pool + ((long)n * nblk + lb) * p.C + (g * CG + tid) * 4) = t;
    }
  }
}

struct LdsCfg { int K, SW, CG, G, TT, TH, nsw, nTt, nTh, nTw; long nblk; size_t lds; };

// Geometry of the LDS-staged kernel for this descriptor; false = not covered (strided, non-square, ...).
static bool lds_cfg(const MspiDwConvDesc* d, LdsCfg& c) {
  static const bool off = getenv("MSPI_DW_LDS") != nullptr && getenv("MSPI_DW_LDS")[0] == '0';   // A/B switch
  if (off || d->kH != d->kW || !(d->kW == 3 || d->kW == 5 || d->kW == 7)) return false;
  if (d->strT != 1 || d->strH != 1 || d->strW != 1 || d->kT > 7) return false;
  // Measured (tools/dw_probe.py, batch 8): staging wins where a thread of the strip kernel would re-fetch the most --
  // 5x5x5 (135 -> 110 us at 8x56^2x64) and 7x7 on maps of 14 or less (48.8 -> 40.7 us at 128x14^2x384) -- and loses on every
  // 3x3x3 shape (27.7 -> 33-47 us at 8x16x14^2x216: the halo of a 3-frame window costs more than L1 re-reads) and on the
  // wide 7x7 maps, where the register-tile kernel already loads each row once.  MSPI_DW_LDS=1 forces it everywhere.
  static const bool all = getenv("MSPI_DW_LDS") != nullptr && getenv("MSPI_DW_LDS")[0] == '1';
  if (!all && !(d->kW == 5 || (d->kW == 7 && d->W <= 14 && d->W > 7))) return false;
  const int To = d->T + 2 * d->padT - d->kT + 1, Ho = d->H + 2 * d->padH - d->kH + 1, Wo = d->W + 2 * d->padW - d->kW + 1;
  if (To <= 0 || Ho <= 0 || Wo <= 0) return false;
  const int CV = d->C / 4;
  c.K = d->kW;
  c.SW = (Wo % 7 == 0) ? 7 : 4;
  const int taps = d->kT * c.K * c.K;
  static const long budget = getenv("MSPI_DW_LDS_KB") ? atol(getenv("MSPI_DW_LDS_KB")) * 1024 : 48 * 1024;
  // exhaustive search over (channel group, frames, rows, strips per row): least staged input per output (the halo factor),
  // subject to the LDS budget and to at least two blocks per CU
  bool found = false;
  double best = 1e30;
  const int nsw_max = (Wo + c.SW - 1) / c.SW;
  for (int cg = 1; cg <= 8; ++cg) {
    if (CV % cg) continue;
    for (int TT = 1; TT <= (d->kT == 1 ? 1 : 8) && TT <= To; ++TT)
      for (int TH = 1; TH <= Ho && TH <= 28; ++TH)
        for (int nsw = 1; nsw <= nsw_max && nsw <= 8; ++nsw) {
          const int TWp = nsw * c.SW;
          const long lds = ((long)(TT + d->kT - 1) * (TH + c.K - 1) * (TWp + c.K - 1) * cg + (long)taps * cg + 256) * 16;
          if (lds > budget) continue;
          const long nTt = (To + TT - 1) / TT, nTh = (Ho + TH - 1) / TH, nTw = (Wo + TWp - 1) / TWp;
          const long blocks = nTt * nTh * nTw * (CV / cg) * d->N;
          // staged elements per useful output (tiles that overhang the map count as waste too)
          double rho = (double)(nTt * (TT + d->kT - 1)) * (nTh * (TH + c.K - 1)) * (nTw * (TWp + c.K - 1)) / ((double)To * Ho * Wo);
          if (blocks < 512) rho *= 4.0;                   // starving the chip is worse than any halo
          rho *= 1.0 + 0.25 / cg;                         // prefer wide channel groups (longer contiguous runs per pixel)
          const long strips = (long)TT * TH * nsw;
          if (strips * cg < 192) rho *= 2.0;              // most of the block's threads must have a strip
          if (rho < best) {
            best = rho; found = true;
            c.CG = cg; c.G = CV / cg; c.TT = TT; c.TH = TH; c.nsw = nsw; c.lds = (size_t)lds;
            c.nTt = (int)nTt; c.nTh = (int)nTh; c.nTw = (int)nTw; c.nblk = nTt * nTh * nTw;
          }
        }
  }
  return found && c.nblk < (1L << 31) && c.G < 65536 && d->N < 65536;
}

struct TileCfg { int K, STR, SW, SH, CG, G, PB, HS, S; long nblk; };

// The tile kernel covers square (k, K, K) kernels, K in {3, 7}, equal H/W stride 1 (or 2 for K = 3).
static bool tile_cfg(const MspiDwConvDesc* d, TileCfg& c) {
  static const bool off = getenv("MSPI_DW_STRIP") != nullptr;   // A/B switch: the older strip kernel
  if (off || d->kH != d->kW || d->strH != d->strW) return false;
  if (!((d->kW == 3 && (d->strW == 1 || d->strW == 2)) || (d->kW == 7 && d->strW == 1))) return false;
  static const bool all = getenv("MSPI_DW_TILE_ALL") != nullptr;
  // measured (profiles/r01): the tile kernel wins on 7x7 with maps >= 28 wide (172 -> 134 us at 56^2 x 96), the strip kernel on
  // the small 7x7 maps, where one round of blocks makes latency, not traffic, the cost.  X3D's 3x3x3 (round 3, r03_dw3_tile.txt):
  // alone the two are equal (X3D-L path 4.34 vs 4.37 ms), but the tile kernel issues a third of the strip kernel's vector loads
  // (weights from LDS, rows shared by two output rows), so beside other kernels -- the way the model runs -- it wins: 0.437 vs
  // 0.409 of HBM peak with four batches in flight, bench line +1.6 %.  MSPI_DW_TILE3=0: strip kernel for 3x3 (A/B).
  static const bool tile3 = !(getenv("MSPI_DW_TILE3") && atoi(getenv("MSPI_DW_TILE3")) == 0);
  if (!all && !(d->kW == 7 && d->W >= 28) && !(d->kW == 3 && tile3)) return false;
  const int Ho = (d->H + 2 * d->padH - d->kH) / d->strH + 1, Wo = (d->W + 2 * d->padW - d->kW) / d->strW + 1;
  const int To = (d->T + 2 * d->padT - d->kT) / d->strT + 1;
  c.K = d->kW; c.STR = d->strW;
  c.SW = (Wo % 7 == 0) ? 7 : 4;
  c.SH = 2;
  const int CV = d->C / 4;
  int g = 1;
  while (CV % g != 0 || CV / g > 32) ++g;
  c.G = g; c.CG = CV / g; c.PB = 256 / c.CG;
  c.HS = (Ho + c.SH - 1) / c.SH; c.S = (Wo + c.SW - 1) / c.SW;
  c.nblk = ((long)To * c.HS * c.S + c.PB - 1) / c.PB;
  return c.nblk < (1L << 31) && c.G < 65536 && d->N < 65536;
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
  static const int unit_xcd = getenv("MSPI_DW_UNIT_XCD") ? atoi(getenv("MSPI_DW_UNIT_XCD")) : 1;
  a.unit_xcd = unit_xcd;
  MSPI_REQUIRE(a.per_sample < (1L << 31) - 256 && d->N < 65536, "%s: more than 2^31 outputs per sample", who);
  return MSPI_OK;
}

}  // namespace mspi

using namespace mspi;

// which strip instantiation serves this descriptor: 0 = (kW 3, stride 1), 1 = (3, 2), 2 = (7, 1), 3 = (5, 1: UniFormer's
// 5x5x5 local "attention"), -1 = generic kernel
static int strip_variant(const MspiDwConvDesc* d) {
  if (d->kH != d->kW) return -1;
  if (d->kW == 3 && d->strW == 1) return 0;
  if (d->kW == 3 && d->strW == 2) return 1;
  if (d->kW == 7 && d->strW == 1) return 2;
  if (d->kW == 5 && d->strW == 1) return 3;
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
  MSPI_REQUIRE(!pool || strip >= 0, "mspi_dwconv_fwd: SE pooling needs a (k,3,3)/(k,5,5)/(k,7,7) kernel with W-stride 1 or 2");
  TileCfg tc;
  LdsCfg lc;
  if (lds_cfg(d, lc)) {
    const dim3 grid((unsigned)lc.nblk, (unsigned)lc.G, (unsigned)a.N);
    const int RW = lc.nsw * lc.SW + lc.K - 1, RH = lc.TH + lc.K - 1;
    const float icg = 1.f / (float)lc.CG, irw = 1.f / (float)RW, irh = 1.f / (float)RH;
#define MSPI_DWL(KK, SWW)                                                                                              \
    do {                                                                                                               \
      if (pool) hipLaunchKernelGGL((dw_lds_kernel<KK, SWW, true>), grid, dim3(256), lc.lds, s, a, lc.CG, lc.TT, lc.TH, lc.nsw, lc.nTh, lc.nTw, icg, irw, irh);  \
      else hipLaunchKernelGGL((dw_lds_kernel<KK, SWW, false>), grid, dim3(256), lc.lds, s, a, lc.CG, lc.TT, lc.TH, lc.nsw, lc.nTh, lc.nTw, icg, irw, irh);      \
    } while (0)
    if (lc.K == 3) { if (lc.SW == 7) MSPI_DWL(3, 7); else MSPI_DWL(3, 4); }
    else if (lc.K == 5) { if (lc.SW == 7) MSPI_DWL(5, 7); else MSPI_DWL(5, 4); }
    else { if (lc.SW == 7) MSPI_DWL(7, 7); else MSPI_DWL(7, 4); }
#undef MSPI_DWL
  } else if (tile_cfg(d, tc)) {
    const dim3 grid((unsigned)tc.nblk, (unsigned)tc.G, (unsigned)a.N);
    const size_t lds = ((size_t)a.kT * tc.K * tc.K * tc.CG + (pool ? 256 : 0)) * sizeof(float4);
#define MSPI_DWT(KK, ST, SWW)                                                                                            \
    do {                                                                                                                 \
      if (pool) hipLaunchKernelGGL((dw_tile_kernel<KK, ST, SWW, 2, true>), grid, dim3(256), lds, s, a, tc.CG, tc.PB, tc.HS, tc.S);  \
      else hipLaunchKernelGGL((dw_tile_kernel<KK, ST, SWW, 2, false>), grid, dim3(256), lds, s, a, tc.CG, tc.PB, tc.HS, tc.S);      \
    } while (0)
    if (tc.K == 7) { if (tc.SW == 7) MSPI_DWT(7, 1, 7); else MSPI_DWT(7, 1, 4); }
    else if (tc.STR == 1) { if (tc.SW == 7) MSPI_DWT(3, 1, 7); else MSPI_DWT(3, 1, 4); }
    else { if (tc.SW == 7) MSPI_DWT(3, 2, 7); else MSPI_DWT(3, 2, 4); }
#undef MSPI_DWT
  } else if (strip >= 0) {
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
      case 3: MSPI_DW_LAUNCH(5, 1); break;
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
  LdsCfg lc;
  if (lds_cfg(d, lc)) return (int)lc.nblk;
  TileCfg tc;
  if (tile_cfg(d, tc)) return (int)tc.nblk;
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
