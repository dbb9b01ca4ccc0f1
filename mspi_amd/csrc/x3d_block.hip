// X3D block, first half, in ONE launch:  u = [swish]( b_bn( dw3x3x3( relu( a_bn( a(x) ) ) ) ) )  (+ squeeze-excite partial sums)
//
// Reference: X3DTransform.forward, SlowFast/resnet_helper.py:296-327 -- `a` 1x1x1 conv dim_in -> dim_inner (2.25x wider),
// a_bn, ReLU, `b` channel-wise 3x3x3 conv, b_bn, [SE on even blocks], Swish.  Unfused (round 1: thin GEMM + depthwise
// kernel) the 2.25x-wide tensor is written once and read once between the two, and each of the 55 blocks is two dependent
// launches of 15-30 us that are latency-, not bandwidth-bound.  Here the expanded tensor never leaves the CU:
//
//   workgroup = (sample, 7 x TW spatial tile, 32-channel chunk of dim_inner, segment of TSEG output frames)
//   march along T with a ring of three frames of `a` outputs in LDS:
//     step i:  a-GEMM of frame t0-1+i on the tile + 1-pixel halo (cells outside the frame / clip are the conv's zero
//              padding) -> ring[i % 3]   : f16x3 split products on v_mfma_f32_16x16x32_f16, weights as the A operand
//              (fragment order, in LDS once per workgroup), the cells' x rows as the B operand straight from global/L2
//              depthwise 3x3x3 of output frame t0+i-2 from the three ring slots: fp32 FMAs, each thread owns 4 channels
//              (its 27 x 4 weights stay in registers) and a strip of SL outputs along W (sliding window: SL+2 LDS reads
//              per kernel row instead of 3*SL)
//   x is re-read per channel chunk from L2 (it is the NARROW tensor); the halo costs (TH+2)(TW+2)/(TH*TW) x (TSEG+2)/TSEG
//   of `a` recomputation, which is MFMA work the thin layers have to spare.
// Bitwise reproducible: no atomics; SE partial sums are one row per workgroup, reduced in a fixed order by mspi_se_gate.
#include "conv_common.h"
#include <stdlib.h>

namespace mspi {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h_ __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_;

struct X3dAbArgs {
  const float* x; const unsigned char* wa; const float* ba; const float* wb; const float* bb; float* u; float* pool;
  int N, T, H, W, Cin, Cmid;
  long ldx, ldu;
  int nch, tiles_w, tiles, nseg, tseg;
  int act;
  float inv_s;
  int* status;
  int dbg;      // ablation switches for tools/x3d_ab_bench.py (MSPI_X3D_DBG): 1 skip the GEMM phase, 2 skip the depthwise phase, 4 no x loads; 0 in production
};

constexpr int X3D_PITCH = 36;   // floats per cell in the ring: 32 channels + 4 pad (conflict-free 16-B stores of 8 consecutive cells)

template <int KS, int TH, int TW, int SL>
__global__ __launch_bounds__(256, 2) void x3d_ab_kernel(const X3dAbArgs p) {
  constexpr int CH = TH + 2, CW = TW + 2, NCELL = CH * CW, NBT = (NCELL + 15) / 16;
  constexpr int WA_BYTES = KS * 4096;                       // [ks][A tile 0/1][hi, lo][lane][8 halves]
  constexpr int SLOT = NCELL * X3D_PITCH;                   // floats per ring slot
  constexpr int NSTRIP = TH * (TW / SL);
  static_assert(TW % SL == 0, "strip length must divide the tile width");
  __shared__ __attribute__((aligned(16))) unsigned char x3d_smem[WA_BYTES + 3 * SLOT * 4 + 128];   // static: may exceed 64 KB
  float* ring = reinterpret_cast<float*>(x3d_smem + WA_BYTES);
  float* bias_a = ring + 3 * SLOT;                          // 32 floats

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // consecutive logical ids share an XCD (hardware deals workgroup b to XCD b % 8): the nch workgroups that re-read one x
  // tile, one per channel chunk, then hit the same 4 MB L2 instead of fetching eight copies through the Infinity Cache
  const int lb = xcd_logical_block((int)blockIdx.x, (int)gridDim.x);
  const int chunk = lb % p.nch;
  int rest = lb / p.nch;
  const int tile = rest % p.tiles;
  const int seg = rest / p.tiles;
  const int n = blockIdx.y;
  const int h0 = (tile / p.tiles_w) * TH, w0 = (tile % p.tiles_w) * TW;
  const int t0 = seg * p.tseg;
  const int tend = min(p.T, t0 + p.tseg);

  // ---- once per workgroup: this chunk's `a` weights (fragment order) into LDS, biases, this thread's depthwise weights
  for (int i = wave; i < WA_BYTES / 1024; i += 4)
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(p.wa + (long)chunk * WA_BYTES + (long)i * 1024 + lane * 16),
                                     (lds_void_*)(x3d_smem + i * 1024), 16, 0, 0);
  if (tid < 32) bias_a[tid] = (chunk * 32 + tid < p.Cmid) ? p.ba[chunk * 32 + tid] : 0.f;
  const int q = tid & 7;                                    // channel quad of the depthwise phase
  const int cq = chunk * 32 + q * 4;
  const bool cok = cq < p.Cmid;
  float4 wreg[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wreg[k] = cok ? *reinterpret_cast<const float4*>(p.wb + (long)k * p.Cmid + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 bq = cok ? *reinterpret_cast<const float4*>(p.bb + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
  bool bad = false;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int li = lane & 15, kg = lane >> 4;
  const int nstep = (tend - t0) + 2;
#pragma unroll 1
  for (int i = 0; i < nstep; ++i) {
    const int ta = t0 - 1 + i;
    float* slot = ring + (i % 3) * SLOT;
    if (p.dbg & 1) {
    } else if (ta < 0 || ta >= p.T) {                       // temporal zero padding of the depthwise conv
      for (int e = tid * 4; e < SLOT; e += 1024) *reinterpret_cast<float4*>(slot + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float* xf = p.x + (((long)n * p.T + ta) * p.H) * (long)p.W * p.ldx;
      constexpr int MAXT = (NBT + 3) / 4;                   // B tiles (16 cells) per wave
      float4 raw[MAXT][KS][2];
      bool ins[MAXT];
      // every load of the step goes out before the first use: one exposed L2 latency per step instead of one per tile
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int cell = (wave + 4 * j) * 16 + li;
        const int ch = cell / CW, cw = cell - ch * CW;
        const int h = h0 - 1 + ch, w = w0 - 1 + cw;
        ins[j] = cell < NCELL && h >= 0 && h < p.H && w >= 0 && w < p.W;
        const float* xr = xf + (ins[j] ? ((long)h * p.W + w) * p.ldx : 0) + 8 * kg;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bool kok = ins[j] && (32 * s + 8 * kg) < p.Cin && !(p.dbg & 4);
          raw[j][s][0] = *reinterpret_cast<const float4*>(kok ? xr + 32 * s : xf);
          raw[j][s][1] = *reinterpret_cast<const float4*>(kok ? xr + 32 * s + 4 : xf);
        }
      }
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int bt = wave + 4 * j;
        if (bt >= NBT) break;
        const int cell = bt * 16 + li;
        const bool inside = ins[j];
        v8h_ xh[KS], xl[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bool kok = inside && (32 * s + 8 * kg) < p.Cin;
          const float4 a = raw[j][s][0], b = raw[j][s][1];
          const float v8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            _Float16 hh, ll;
            split_f16(kok ? v8[e] : 0.f, hh, ll);
            xh[s][e] = hh; xl[s][e] = ll;
          }
        }
#pragma unroll
        for (int at = 0; at < 2; ++at) {
          v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const unsigned char* wp = x3d_smem + ((s * 2 + at) * 2) * 1024 + lane * 16;
            const v8h_ wh = *reinterpret_cast<const v8h_*>(wp);
            const v8h_ wl = *reinterpret_cast<const v8h_*>(wp + 1024);
            if (!kSingleProduct) {
              acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[s], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[s], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[s], acc, 0, 0, 0);
          }
          // lane (li, kg): cell li of this B tile, channels at*16 + 4*kg + 0..3;  a_bn bias, ReLU, zero outside the frame
          const float4 bv = *reinterpret_cast<const float4*>(bias_a + at * 16 + 4 * kg);
          bad |= inside && (nonfinite(acc[0]) | nonfinite(acc[1]) | nonfinite(acc[2]) | nonfinite(acc[3]));
          float4 o;
          o.x = inside ? fmaxf(fmaf(acc[0], p.inv_s, bv.x), 0.f) : 0.f;
          o.y = inside ? fmaxf(fmaf(acc[1], p.inv_s, bv.y), 0.f) : 0.f;
          o.z = inside ? fmaxf(fmaf(acc[2], p.inv_s, bv.z), 0.f) : 0.f;
          o.w = inside ? fmaxf(fmaf(acc[3], p.inv_s, bv.w), 0.f) : 0.f;
          if (cell < NCELL) *reinterpret_cast<float4*>(slot + cell * X3D_PITCH + at * 16 + 4 * kg) = o;
        }
      }
    }
    __syncthreads();
    if (i >= 2 && !(p.dbg & 2)) {
      const int to = t0 + i - 2;                            // output frame; its inputs: frames to-1, to, to+1 = steps i-2, i-1, i
      const float* s0 = ring + ((i - 2) % 3) * SLOT;
      const float* s1 = ring + ((i - 1) % 3) * SLOT;
      const float* s2 = ring + (i % 3) * SLOT;
#pragma unroll 1
      for (int it = tid >> 3; it < NSTRIP; it += 32) {
        const int hh = it / (TW / SL), ws = (it - hh * (TW / SL)) * SL;
        float4 acc[SL];
#pragma unroll
        for (int j = 0; j < SL; ++j) acc[j] = bq;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
          asm volatile("" ::: "memory");                    // keep the LDS reads of one tap plane (12 x 16 B) in flight, not all 36
          const float* sl = dt == 0 ? s0 : (dt == 1 ? s1 : s2);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const float* row = sl + ((hh + kh) * CW + ws) * X3D_PITCH + q * 4;
            float4 win[SL + 2];
#pragma unroll
            for (int c = 0; c < SL + 2; ++c) win[c] = *reinterpret_cast<const float4*>(row + c * X3D_PITCH);
#pragma unroll
            for (int j = 0; j < SL; ++j)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) {
                const float4 wv = wreg[(dt * 3 + kh) * 3 + kw];
                acc[j].x = fmaf(win[j + kw].x, wv.x, acc[j].x);
                acc[j].y = fmaf(win[j + kw].y, wv.y, acc[j].y);
                acc[j].z = fmaf(win[j + kw].z, wv.z, acc[j].z);
                acc[j].w = fmaf(win[j + kw].w, wv.w, acc[j].w);
              }
          }
        }
        float* up = p.u + ((((long)n * p.T + to) * p.H + (h0 + hh)) * (long)p.W + (w0 + ws)) * p.ldu + cq;
#pragma unroll
        for (int j = 0; j < SL; ++j) {
          float4 v = acc[j];
          psum.x += v.x; psum.y += v.y; psum.z += v.z; psum.w += v.w;
          if (p.act == MSPI_ACT_SWISH) {
            v.x = fast_swish(v.x); v.y = fast_swish(v.y); v.z = fast_swish(v.z); v.w = fast_swish(v.w);
          }
          if (cok) *reinterpret_cast<float4*>(up + (long)j * p.ldu) = v;
        }
      }
    }
    __syncthreads();
  }
  report_nonfinite(p.status, bad);
  if (p.pool) {     // squeeze-excite partial sums of the pre-activation output: one row per workgroup, fixed order
    float* red = ring;                                      // all ring reads are behind the loop's last barrier
    *reinterpret_cast<float4*>(red + tid * 4) = psum;
    __syncthreads();
    if (tid < 32) {
      const int qq = tid >> 2, comp = tid & 3;
      float s = 0.f;
      for (int r = 0; r < 32; ++r) s += red[(r * 8 + qq) * 4 + comp];
      const int c = chunk * 32 + tid;
      if (c < p.Cmid) p.pool[((long)n * (p.tiles * p.nseg) + (seg * p.tiles + tile)) * p.Cmid + c] = s;
    }
  }
}

template <int KS, int TH, int TW, int SL>
static void launch_ab(const X3dAbArgs& a, hipStream_t s) {
  const dim3 grid((unsigned)(a.nch * a.tiles * a.nseg), (unsigned)a.N);
  hipLaunchKernelGGL((x3d_ab_kernel<KS, TH, TW, SL>), grid, dim3(256), 0, s, a);
}

static int x3d_ks(int cin) { return (cin + 31) / 32; }

// frames per T segment: as long as possible (less `a` recomputation) while the grid still covers the chip about twice
static int x3d_tseg(int N, int T, int tiles, int nch) {
  int best = T;
  for (int ts = T; ts >= 2; ts = (ts + 1) / 2) {
    best = ts;
    if ((long)N * tiles * nch * ((T + ts - 1) / ts) >= 448) break;
    if (ts == 2) break;
  }
  return best;
}

}  // namespace mspi

using namespace mspi;

extern "C" int mspi_x3d_ab_supported(const MspiX3dAbDesc* d) {
  if (!d) return 0;
  const int ks = x3d_ks(d->Cin);
  const bool k_ok = ks == 1 || ks == 2 || ks == 3 || (ks == 6 && d->W == 7);   // K > 96 on 14-wide tiles: over the register budget
  const bool tile_ok = d->H % 7 == 0 && (d->W % 14 == 0 || d->W == 7);
  return k_ok && tile_ok && d->Cin % 8 == 0 && d->Cmid % 4 == 0 && d->Cin >= 8 && d->Cmid >= 4 && d->T >= 1;
}

static void x3d_geometry(const MspiX3dAbDesc* d, X3dAbArgs& a) {
  const int tw = d->W % 14 == 0 ? 14 : 7;
  a.nch = (d->Cmid + 31) / 32;
  a.tiles_w = d->W / tw;
  a.tiles = (d->H / 7) * a.tiles_w;
  a.tseg = x3d_tseg(d->N, d->T, a.tiles, a.nch);
  a.nseg = (d->T + a.tseg - 1) / a.tseg;
}

extern "C" int mspi_x3d_ab_pool_rows(const MspiX3dAbDesc* d) {
  if (!mspi_x3d_ab_supported(d)) return 0;
  X3dAbArgs a;
  x3d_geometry(d, a);
  return a.tiles * a.nseg;
}

extern "C" size_t mspi_x3d_ab_packed_bytes(int32_t Cin, int32_t Cmid) {
  return (size_t)((Cmid + 31) / 32) * x3d_ks(Cin) * 4096;
}

extern "C" int mspi_x3d_ab_fwd(const MspiX3dAbDesc* d, const void* x, const void* wa_packed, const void* bias_a, const void* wb,
                               const void* bias_b, void* u, void* pool, void* stream) {
  MSPI_REQUIRE(d && x && wa_packed && bias_a && wb && bias_b && u, "mspi_x3d_ab_fwd: null argument");
  MSPI_REQUIRE(mspi_x3d_ab_supported(d), "mspi_x3d_ab_fwd: shape N=%d T=%d H=%d W=%d Cin=%d Cmid=%d is outside the fused kernel's range",
               d->N, d->T, d->H, d->W, d->Cin, d->Cmid);
  MSPI_REQUIRE(d->N >= 1 && d->N < 65536 && d->ldx >= d->Cin && d->ldu >= d->Cmid && d->ldx % 4 == 0 && d->ldu % 4 == 0,
               "mspi_x3d_ab_fwd: row strides must cover the row and be multiples of 4 floats");
  MSPI_REQUIRE(aligned16(x) && aligned16(u) && aligned16(wb) && aligned16(bias_b) && aligned16(wa_packed), "mspi_x3d_ab_fwd: 16-byte alignment");
  MSPI_REQUIRE(d->act == MSPI_ACT_NONE || d->act == MSPI_ACT_SWISH, "mspi_x3d_ab_fwd: act must be NONE or SWISH");
  MSPI_REQUIRE(d->wa_scale > 0.f, "mspi_x3d_ab_fwd: wa_scale must be positive");
  X3dAbArgs a;
  a.x = (const float*)x; a.wa = (const unsigned char*)wa_packed; a.ba = (const float*)bias_a; a.wb = (const float*)wb;
  a.bb = (const float*)bias_b; a.u = (float*)u; a.pool = (float*)pool;
  a.N = d->N; a.T = d->T; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cmid = d->Cmid; a.ldx = d->ldx; a.ldu = d->ldu;
  a.act = d->act; a.inv_s = 1.0f / d->wa_scale;
  static const int dbg = getenv("MSPI_X3D_DBG") ? atoi(getenv("MSPI_X3D_DBG")) : 0;
  static const int tseg_env = getenv("MSPI_X3D_TSEG") ? atoi(getenv("MSPI_X3D_TSEG")) : 0;
  a.dbg = dbg;
  a.status = g_status_word;
  x3d_geometry(d, a);
  if (tseg_env > 0 && !pool) { a.tseg = tseg_env; a.nseg = (d->T + a.tseg - 1) / a.tseg; }
  hipStream_t s = (hipStream_t)stream;
  const int ks = x3d_ks(d->Cin);
  if (d->W % 14 == 0) {
    switch (ks) {
      case 1: launch_ab<1, 7, 14, 2>(a, s); break;
      case 2: launch_ab<2, 7, 14, 2>(a, s); break;
      default: launch_ab<3, 7, 14, 2>(a, s); break;
    }
  } else {
    switch (ks) {
      case 1: launch_ab<1, 7, 7, 1>(a, s); break;
      case 2: launch_ab<2, 7, 7, 1>(a, s); break;
      case 3: launch_ab<3, 7, 7, 1>(a, s); break;
      default: launch_ab<6, 7, 7, 1>(a, s); break;
    }
  }
  return check_launch("mspi_x3d_ab_fwd");
}
