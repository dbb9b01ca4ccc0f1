// X3D res-stage, blocks 1..n-1 (all stride 1, dim_in == dim_out), as ONE persistent launch.
//
// Reference: ResStage.forward over ResBlock / X3DTransform (SlowFast/resnet_helper.py:296-351 `a` 1x1x1 -> a_bn -> ReLU -> `b`
// channel-wise 3x3x3 -> b_bn -> [SE :27-73 on even blocks] -> Swish -> `c` 1x1x1 -> c_bn; :593-616 `x + branch2(x)` -> ReLU).
// As separate launches a block of stage 4 (14x14 maps, batch 8) is 3-4 dependent kernels of 17-28 us on 10-20 MB tensors:
// launch + one exposed memory latency after another on 98 positions per CU, 2.6 ms for the 39 blocks of stages 4 and 5.
// Here a stage is one launch:
//
//   * one SAMPLE per XCD group (workgroup b works for group b % 8 -- observed dispatch deals b to XCD b % 8, so a sample's
//     tensors stay in one 4 MB L2 / the Infinity Cache; placement is a speed matter only, see "hand-offs"), P <= 32 workgroups
//     per sample, each owning a tile of TH rows of one frame (R = TH*W <= 128 positions = rows of the [M, C] matrices);
//   * per block three phases on the workgroup's own rows:
//       A  t = relu(a(x))            f16x3 MFMA (32x32x16), B operand = the rows' f16 hi/lo planes, A operand = weight
//                                     fragments streamed through a two-stage LDS ring by LDS-DMA
//       B  u = b(t) (+ pool sums)    fp32 FMAs, register tile of RPI rows x 7 columns x 4 channels per thread, t read through
//                                     L2 with the 1-position halo owned by the neighbouring tiles
//       C  x = relu(c(swish(g u)) + x)   as A; the epilogue writes x as fp32 (residual, stage output) and as planes (next A)
//     everything a workgroup hands to ITSELF (x, u, the planes) goes through plain global stores / loads and stays in L1/L2;
//   * hand-offs between workgroups -- the t halo, the squeeze-excite pool sums -- follow the measured write-through protocol
//     (MI355X_MICROARCH.md "Valid forms", row 1): every handed-off byte is stored `sc1` (16-B buffer stores), every storing
//     wave drains `vmcnt(0)`, the workgroup's barrier, ONE lane publishes (an `sc1` epoch store / an agent-scope atomic add);
//     the consumer polls that word relaxed (`sc1` loads, bounded), joins a workgroup barrier, and EVERY load of handed-off
//     bytes is an `sc1` buffer load to registers.  No fence, nothing depends on dispatch order or XCD placement.
//   * every spin is bounded: on a timeout (workgroups not co-resident) the kernel sets the abort word, every workgroup drains
//     out and the status word reports it -- wrong output and an error, never a hang.
// Bitwise reproducible: no float atomics; pool sums are one row per workgroup, added in a fixed order by every consumer.
#include "common.h"

namespace mspi {

typedef _Float16 v8h_s __attribute__((ext_vector_type(8)));
typedef _Float16 v4h_s __attribute__((ext_vector_type(4)));
typedef float v16f_s __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_s;

struct X3dStageArgs {
  const float* xin; float* y;
  float* tbuf;            // [2][M][Ds]
  float* ubuf;            // [M][Ds]          pre-gate u of squeeze-excite blocks
  _Float16* up;           // [2][M][KU]       swish(g u) as hi / lo planes (KU = 16 KSC)
  _Float16* xp;           // [2][M][C]        x as hi / lo planes
  float* pool;            // [2][N][P][Ds]    pool partial sums, by parity of the SE block index
  unsigned* sync;         // [N*P] t epochs | [N] pool arrivals | abort word (zeroed by the launch function)
  const unsigned char* wq; const float* wf;
  long wq_stride, wf_stride;
  long M;
  int N, T, H, W, C, D, Ds, F;
  int nblocks;
  unsigned se_mask;
  int TH, tiles_f, P;
  int KSA, NA, KSC, NC, KPA, KPC;
  int RPI;                // rows per depthwise item
  int ring_bytes;         // bytes of one LDS ring stage
  int* status;
};

constexpr unsigned X3DS_SPIN_LIMIT = 1u << 21;

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one wave polls up to 64 words (one per lane; lanes with p == nullptr are satisfied); wave-uniform result
__device__ __forceinline__ bool wave_wait_ge(const unsigned* p, unsigned target, unsigned* abortw) {
  for (unsigned it = 0;; ++it) {
    const bool ok = p == nullptr || ld_relaxed(p) >= target;
    if (__all(ok)) return true;
    if ((it & 31u) == 31u && ld_relaxed(abortw) != 0u) return false;
    if (it > X3DS_SPIN_LIMIT) {
      __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

typedef float f32x4_s __attribute__((ext_vector_type(4)));
// NOTE: the loaded vector is re-typed as a WHOLE (bit_cast of the 4-vector).  Picking v[0..3] out of the builtin's result
// element by element makes hipcc (ROCm 7.2) narrow the load to buffer_load_dword and use that one dword for all four.
__device__ __forceinline__ float4 ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4_s v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);     // aux 16 = sc1: L1 bypassed
  const f32x4_s f = __builtin_bit_cast(f32x4_s, v);
  return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
  const f32x4_s f = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_s, f), r, (int)byte_off, 0, 16);   // write-through
}

__device__ __forceinline__ void split4(float4 v, v4h_s& hi, v4h_s& lo) {
  _Float16 h, l;
  split_f16(v.x, h, l); hi[0] = h; lo[0] = l;
  split_f16(v.y, h, l); hi[1] = h; lo[1] = l;
  split_f16(v.z, h, l); hi[2] = h; lo[2] = l;
  split_f16(v.w, h, l); hi[3] = h; lo[3] = l;
}
__device__ __forceinline__ float swish1(float v) { return v / (1.f + __expf(-v)); }

// The kernel is a loop over (sample, block) that calls three NON-INLINED phase functions.  As one inlined body the
// optimiser unswitched the block loop three ways and kept ~380 uniform values live across all phases (hundreds of spilled
// SGPRs and VGPRs); as functions each phase is allocated on its own.  A phase re-derives its geometry from the workgroup id
// and reads the launch parameters straight from the kernarg segment (scalar loads), so nothing but (kernarg pointer, n, k) crosses a call.
typedef const __attribute__((address_space(4))) X3dStageArgs* XsArgs;
// The kernel reads its kernarg segment pointer and hands it to the phases (inside a non-kernel function the builtin folds to
// null); a phase makes it scalar again, so every parameter read is an s_load.
__device__ __forceinline__ XsArgs xs_args(XsArgs kp) {
  const unsigned long long v = (unsigned long long)kp;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (XsArgs)(((unsigned long long)hi << 32) | lo);
}

struct XsGeo {
  int tile, tf, tr, h0, th, R, RGe, WPR, rg, cw;
  long row0, HW;
};
__device__ __forceinline__ XsGeo xs_geo(XsArgs p, int n) {
  XsGeo g;
  const int wave = threadIdx.x >> 6;
  g.tile = blockIdx.x >> 3;
  g.tf = g.tile / p->tiles_f; g.tr = g.tile - g.tf * p->tiles_f;
  g.h0 = g.tr * p->TH;
  g.th = min(p->TH, p->H - g.h0);
  g.R = g.th * p->W;
  const int RG = (g.R + 31) >> 5;
  g.RGe = RG == 3 ? 4 : RG;                 // 1, 2 or 4 row groups of 32
  g.WPR = 4 / g.RGe;                        // waves that share one row group (they split the output chunks)
  g.rg = wave % g.RGe; g.cw = wave / g.RGe;
  g.HW = (long)p->H * p->W;
  g.row0 = ((long)n * p->T + g.tf) * g.HW + (long)g.h0 * p->W;       // first row of the tile in the [M, .] matrices
  return g;
}

// float parameters of block k: ba[NA*32] bc[C] bb[Ds] wb[27*Ds] w1[F*Ds] b1[F] w2[Ds*F] b2[Ds] inv_sa inv_sc
struct XsParams { const float *ba, *bc, *bb, *wb, *w1, *b1, *w2, *b2; };
__device__ __forceinline__ XsParams xs_params(XsArgs p, int k) {
  XsParams f;
  f.ba = p->wf + (long)k * p->wf_stride;
  f.bc = f.ba + p->NA * 32;
  f.bb = f.bc + p->C;
  f.wb = f.bb + p->Ds;
  f.w1 = f.wb + 27 * p->Ds;
  f.b1 = f.w1 + p->F * p->Ds;
  f.w2 = f.b1 + p->F;
  f.b2 = f.w2 + p->Ds * p->F;
  return f;
}

__device__ __forceinline__ unsigned char* xs_ring() {
  extern __shared__ __attribute__((aligned(16))) unsigned char xs_smem[];
  return xs_smem;
}

// ---------------------------------------------------------------------------------------------------------------------
// GEMM phase.  PHASE 0 (A): t = relu(a(x) + ba) -> tbuf[k & 1] (write-through), then the tile's epoch is published.
//              PHASE 1 (C): x = relu(c(u planes) + bc + x) -> y (fp32) and the x planes.
// Wave (rg, cw) accumulates the output chunks cw, cw + WPR, ... (MAXCH of them at most) of row group rg; the B operand is
// the rows' hi / lo planes (plain loads: the workgroup's own data), the A operand the weight fragments of KP k-steps x all
// chunks per LDS ring stage.  Returns 1 when a result was not finite.
template <int MAXCH, int PHASE>
__device__ __attribute__((noinline)) int xs_gemm_phase(XsArgs kp, int n_, int k_, int first_) {
  const int n = __builtin_amdgcn_readfirstlane(n_), k = __builtin_amdgcn_readfirstlane(k_);
  const bool first = __builtin_amdgcn_readfirstlane(first_) != 0;
  const XsArgs p = xs_args(kp);
  const XsGeo g = xs_geo(p, n);
  const XsParams f = xs_params(p, k);
  unsigned char* ring = xs_ring();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int KS = PHASE == 0 ? p->KSA : p->KSC, NCH = PHASE == 0 ? p->NA : p->NC, KP = PHASE == 0 ? p->KPA : p->KPC;
  const int ldp = PHASE == 0 ? p->C : p->KSC * 16;
  const int ring_bytes = p->ring_bytes;
  const unsigned char* wq = p->wq + (long)k * p->wq_stride + (PHASE == 0 ? 0L : (long)p->KSA * p->NA * 2048);
  const long M = p->M;
  const int brow = min(g.rg * 32 + li, g.R - 1);     // rows past the tile repeat its last row; their results are dropped
  const bool rvalid = g.rg * 32 + li < g.R;
  bool bad = false;

  auto dma_stage = [&](const unsigned char* src, int bytes, int slot) {     // 1-KB pieces dealt to the four waves
    unsigned char* dst = ring + slot * ring_bytes;
    for (int i = wave; i < (bytes >> 10); i += 4)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + (long)i * 1024 + lane * 16),
                                       (lds_void_s*)(dst + (long)i * 1024), 16, 0, 0);
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // this phase's planes are complete; the ring is free
  const int stage_bytes = KP * NCH * 2048;
  const int npanel = (KS + KP - 1) / KP;
  dma_stage(wq, min(KP, KS) * NCH * 2048, 0);
  const int mych = max(0, (NCH - g.cw + g.WPR - 1) / g.WPR);
  v16f_s acc[MAXCH];
#pragma unroll
  for (int i = 0; i < MAXCH; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const _Float16* bhi = (PHASE == 0 ? p->xp : p->up) + (g.row0 + brow) * ldp + 8 * lh;
  const _Float16* blo = bhi + M * ldp;
#pragma unroll 1
  for (int pn = 0; pn < npanel; ++pn) {
    const int ks0 = pn * KP;
    const int nks = min(KP, KS - ks0);
    v8h_s fh[4], fl[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ks = min(ks0 + kk, KS - 1);
      fh[kk] = *reinterpret_cast<const v8h_s*>(bhi + 16 * ks);
      fl[kk] = *reinterpret_cast<const v8h_s*>(blo + 16 * ks);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // stage pn has landed for everybody; stage pn-1 is no longer read
    if (pn + 1 < npanel) dma_stage(wq + (long)(pn + 1) * stage_bytes, min(KP, KS - (pn + 1) * KP) * NCH * 2048, (pn + 1) & 1);
    const unsigned char* st = ring + (pn & 1) * ring_bytes + lane * 16;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < nks) {
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
          if (i < mych) {
            const int j = g.cw + g.WPR * i;
            const v8h_s wh = *reinterpret_cast<const v8h_s*>(st + ((kk * NCH + j) * 2 + 0) * 1024);
            const v8h_s wl = *reinterpret_cast<const v8h_s*>(st + ((kk * NCH + j) * 2 + 1) * 1024);
            if (!kSingleProduct) {
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fh[kk], acc[i], 0, 0, 0);
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fl[kk], acc[i], 0, 0, 0);
            }
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fh[kk], acc[i], 0, 0, 0);
          }
        }
      }
    }
  }
  // epilogue: lane (li, lh) holds, for row rg*32 + li, columns j*32 + q*8 + 4*lh + 0..3
  if (PHASE == 0) {
    const float inv_s = f.b2[p->Ds];
    const int Ds = p->Ds;
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(p->tbuf + (long)(k & 1) * M * Ds, 0, (int)(M * Ds * 4), 0x00020000);
    const unsigned trow = (unsigned)((g.row0 + g.rg * 32 + li) * Ds) * 4u;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      if (i < mych) {
        const int j = g.cw + g.WPR * i;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = j * 32 + q * 8 + 4 * lh;
          const float4 b = *reinterpret_cast<const float4*>(f.ba + c);
          float4 v;
          v.x = fmaf(acc[i][q * 4 + 0], inv_s, b.x); v.y = fmaf(acc[i][q * 4 + 1], inv_s, b.y);
          v.z = fmaf(acc[i][q * 4 + 2], inv_s, b.z); v.w = fmaf(acc[i][q * 4 + 3], inv_s, b.w);
          bad |= rvalid && (nonfinite(v.x) | nonfinite(v.y) | nonfinite(v.z) | nonfinite(v.w));
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          if (rvalid && c < Ds) st_sc1(rs_t, trow + (unsigned)c * 4u, v);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // EVERY storing wave drains before the barrier ...
    __syncthreads();
    if (tid == 0)                                            // ... and ONE lane publishes the tile's epoch
      __hip_atomic_store(p->sync + (long)n * p->P + g.tile, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    const float inv_s = f.b2[p->Ds + 1];
    const int C = p->C;
    const float* xcur = first ? p->xin : p->y;
    float* y = p->y;
    _Float16* xp = p->xp;
    const long xrow = (g.row0 + g.rg * 32 + li) * C;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      if (i < mych) {
        const int j = g.cw + g.WPR * i;
        float4 rv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rv[q] = rvalid ? *reinterpret_cast<const float4*>(xcur + xrow + j * 32 + q * 8 + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = j * 32 + q * 8 + 4 * lh;
          const float4 b = *reinterpret_cast<const float4*>(f.bc + c);
          float4 v;
          v.x = fmaf(acc[i][q * 4 + 0], inv_s, b.x) + rv[q].x; v.y = fmaf(acc[i][q * 4 + 1], inv_s, b.y) + rv[q].y;
          v.z = fmaf(acc[i][q * 4 + 2], inv_s, b.z) + rv[q].z; v.w = fmaf(acc[i][q * 4 + 3], inv_s, b.w) + rv[q].w;
          bad |= rvalid && (nonfinite(v.x) | nonfinite(v.y) | nonfinite(v.z) | nonfinite(v.w));
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          if (rvalid) {
            *reinterpret_cast<float4*>(y + xrow + c) = v;
            v4h_s hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<v4h_s*>(xp + xrow + c) = hi;
            *reinterpret_cast<v4h_s*>(xp + M * C + xrow + c) = lo;
          }
        }
      }
    }
  }
  return bad ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Depthwise phase: waits for the neighbouring tiles' t, then u = b(t) + bb on the tile.  Thread = one item of RPI rows x one
// strip (<= 7 columns) x 4 channels; a 3 x (RPI+2) x 9 window of t slides through registers.  Blocks without squeeze-excite
// write swish(u) as planes; SE blocks write fp32 u and the item's pool sums (LDS `red`).  Returns 0, or 1 on a timeout.
template <int RPI>
__device__ __attribute__((noinline)) int xs_dw_phase(XsArgs kp, int n_, int k_) {
  const int n = __builtin_amdgcn_readfirstlane(n_), k = __builtin_amdgcn_readfirstlane(k_);
  const XsArgs p = xs_args(kp);
  const XsGeo g = xs_geo(p, n);
  const XsParams f = xs_params(p, k);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ds = p->Ds, W = p->W, H = p->H, T = p->T;
  const long M = p->M;
  float* red = reinterpret_cast<float*>(xs_ring() + 2 * p->ring_bytes);
  int* sflag = reinterpret_cast<int*>(red + 256 * 4 + 2 * Ds + p->F);
  const bool se = (p->se_mask >> k) & 1u;

  // ---- wait for the tiles whose t this tile's halo reads (epoch k + 1), one lane per neighbour
  if (wave == 0) {
    const unsigned* w = nullptr;
    if (lane < 9) {
      const int dt = lane / 3 - 1, dr = lane % 3 - 1;
      const int t2 = g.tf + dt, r2 = g.tr + dr;
      if ((dt != 0 || dr != 0) && t2 >= 0 && t2 < T && r2 >= 0 && r2 < p->tiles_f) w = p->sync + (long)n * p->P + t2 * p->tiles_f + r2;
    }
    const bool ok = wave_wait_ge(w, (unsigned)(k + 1), p->sync + (long)p->N * p->P + p->N);
    if (lane == 0) *sflag = ok ? 1 : 0;
  }
  __syncthreads();
  if (*sflag == 0) return 1;

  const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(p->tbuf + (long)(k & 1) * M * Ds, 0, (int)(M * Ds * 4), 0x00020000);
  const int NQ = Ds >> 2;
  const int nst = (W + 6) / 7;
  const int SW = (W + nst - 1) / nst;                // strip width <= 7
  const int nrp = (g.th + RPI - 1) / RPI;
  const int nitems = nrp * nst * NQ;
  const int KU = p->KSC * 16;
  if (tid < nitems) {
    const int it = tid;
    const int q = it % NQ;
    const int rest = it / NQ;
    const int s = rest % nst, rp = rest / nst;
    const int w0 = s * SW;
    const int wn = min(SW, W - w0);                  // columns of this strip
    const int r0 = rp * RPI;
    const int rn = min(RPI, g.th - r0);
    float4 acc[RPI][7];
    const float4 bq = *reinterpret_cast<const float4*>(f.bb + 4 * q);
#pragma unroll
    for (int a = 0; a < RPI; ++a)
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[a][j] = bq;
    // Zero padding costs no select: a buffer load whose offset is past num_records returns zeros, so an invalid row / column /
    // frame puts XS_OOB into the offset (num_records < 2^30 is checked on the host; two of them still stay out of range).
    constexpr unsigned XS_OOB = 0x40000000u;
    const unsigned ds4 = (unsigned)Ds * 4u;
    unsigned coff[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int ww = w0 + c - 1;
      coff[c] = (ww >= 0 && ww < W) ? (unsigned)c * ds4 : XS_OOB;
    }
#pragma unroll 1
    for (int dt = 0; dt < 3; ++dt) {
      const int tt = g.tf + dt - 1;
      const bool tok = tt >= 0 && tt < T;
      float4 wv[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) wv[e] = *reinterpret_cast<const float4*>(f.wb + (long)(dt * 9 + e) * Ds + 4 * q);
      // byte offset of (frame tt, row h0 + r0 - 1, column w0 - 1, channel 4q); rows advance by W * Ds * 4
      const long cell0 = ((long)n * T + (tok ? tt : g.tf)) * g.HW + (long)(g.h0 + r0 - 1) * W + (w0 - 1);
      const unsigned base = (unsigned)(int)(cell0 * (long)ds4 + 16 * q);     // may wrap for the halo row above the first frame: masked
#pragma unroll
      for (int ir = 0; ir < RPI + 2; ++ir) {
        const int hh = g.h0 + r0 + ir - 1;
        const bool hok = tok && hh >= 0 && hh < H && ir <= rn + 1;
        const unsigned rowoff = base + (unsigned)ir * (unsigned)W * ds4;
        float4 win[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) win[c] = ld_sc1(rs_t, hok ? rowoff + coff[c] : 2u * XS_OOB);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int a = ir - kh;                     // output row of the item fed by input row ir through tap row kh
          if (a >= 0 && a < RPI) {
#pragma unroll
            for (int j = 0; j < 7; ++j)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) {
                const float4 w4 = wv[kh * 3 + kw];
                acc[a][j].x = fmaf(win[j + kw].x, w4.x, acc[a][j].x);
                acc[a][j].y = fmaf(win[j + kw].y, w4.y, acc[a][j].y);
                acc[a][j].z = fmaf(win[j + kw].z, w4.z, acc[a][j].z);
                acc[a][j].w = fmaf(win[j + kw].w, w4.w, acc[a][j].w);
              }
          }
        }
        __builtin_amdgcn_sched_barrier(0);           // one input row's nine loads in flight, not the whole frame's
      }
    }
    float4 ps = make_float4(0.f, 0.f, 0.f, 0.f);
    float* ubuf = p->ubuf;
    _Float16* up = p->up;
#pragma unroll
    for (int a = 0; a < RPI; ++a)
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        if (a < rn && j < wn) {
          const long row = g.row0 + (long)(r0 + a) * W + w0 + j;
          float4 v = acc[a][j];
          if (se) {
            ps.x += v.x; ps.y += v.y; ps.z += v.z; ps.w += v.w;
            *reinterpret_cast<float4*>(ubuf + row * Ds + 4 * q) = v;
          } else {
            v.x = swish1(v.x); v.y = swish1(v.y); v.z = swish1(v.z); v.w = swish1(v.w);
            v4h_s hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<v4h_s*>(up + row * KU + 4 * q) = hi;
            *reinterpret_cast<v4h_s*>(up + M * KU + row * KU + 4 * q) = lo;
          }
        }
      }
    if (se) *reinterpret_cast<float4*>(red + it * 4) = ps;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Squeeze-excite phase of an SE block: the tile's pool sums -> one write-through row, arrive, wait for the sample's P rows,
// gate = sigmoid(fc2(relu(fc1(mean)))) (every workgroup adds the P rows in the same order), then u -> swish(g u) as planes.
// Returns 0, or 1 on a timeout.
__device__ __attribute__((noinline)) int xs_se_phase(XsArgs kp, int n_, int k_, int se_idx_) {
  const int n = __builtin_amdgcn_readfirstlane(n_), k = __builtin_amdgcn_readfirstlane(k_);
  const int se_idx = __builtin_amdgcn_readfirstlane(se_idx_);
  const XsArgs p = xs_args(kp);
  const XsGeo g = xs_geo(p, n);
  const XsParams f = xs_params(p, k);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ds = p->Ds, F = p->F, P = p->P;
  const long M = p->M;
  float* red = reinterpret_cast<float*>(xs_ring() + 2 * p->ring_bytes);
  float* smean = red + 256 * 4;
  float* sgate = smean + Ds;
  float* shid = sgate + Ds;
  int* sflag = reinterpret_cast<int*>(shid + F);
  const int NQ = Ds >> 2;
  const int RPI = p->RPI;
  const int nst = (p->W + 6) / 7;
  const int nrp = (g.th + RPI - 1) / RPI;
  unsigned* pcount = p->sync + (long)p->N * P + n;

  __syncthreads();                                   // the items' pool sums are in `red`
  float* prow = p->pool + (((long)(se_idx & 1) * p->N + n) * P) * Ds;
  const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(prow, 0, P * Ds * 4, 0x00020000);
  if (tid < NQ) {
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < nrp * nst; ++r) {
      const float4 v = *reinterpret_cast<const float4*>(red + (r * NQ + tid) * 4);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    st_sc1(rs_p, (unsigned)(g.tile * Ds + 4 * tid) * 4u, s4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the storing waves drain (the u stores of the depthwise phase too)
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(pcount, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (wave == 0) {
    const bool ok = wave_wait_ge(lane == 0 ? pcount : nullptr, (unsigned)(P * (se_idx + 1)), p->sync + (long)p->N * P + p->N);
    if (lane == 0) *sflag = ok ? 1 : 0;
  }
  __syncthreads();
  if (*sflag == 0) return 1;
  if (tid < NQ) {
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < P; ++r) {
      const float4 v = ld_sc1(rs_p, (unsigned)(r * Ds + 4 * tid) * 4u);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    const float inv = 1.f / (float)((long)p->T * g.HW);
    smean[4 * tid + 0] = s4.x * inv; smean[4 * tid + 1] = s4.y * inv;
    smean[4 * tid + 2] = s4.z * inv; smean[4 * tid + 3] = s4.w * inv;
  }
  __syncthreads();
  {   // fc1 + ReLU: F <= 32 hidden units, 8 lanes each
    const int hf = tid >> 3, part = tid & 7;
    float s = 0.f;
    if (hf < F)
      for (int c = part; c < Ds; c += 8) s = fmaf(f.w1[(long)hf * Ds + c], smean[c], s);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (hf < F && part == 0) shid[hf] = fmaxf(s + f.b1[hf], 0.f);
  }
  __syncthreads();
  for (int c = tid; c < Ds; c += 256) {
    float s = f.b2[c];
    for (int j = 0; j < F; ++j) s = fmaf(f.w2[(long)c * F + j], shid[j], s);
    sgate[c] = 1.f / (1.f + __expf(-s));
  }
  __syncthreads();
  const int KU = p->KSC * 16;
  const float* ubuf = p->ubuf;
  _Float16* up = p->up;
  for (int i = tid; i < g.R * NQ; i += 256) {
    const int r = i / NQ, q = i - r * NQ;
    float4 v = *reinterpret_cast<const float4*>(ubuf + (g.row0 + r) * Ds + 4 * q);
    const float4 gq = *reinterpret_cast<const float4*>(sgate + 4 * q);
    v.x = swish1(v.x * gq.x); v.y = swish1(v.y * gq.y); v.z = swish1(v.z * gq.z); v.w = swish1(v.w * gq.w);
    v4h_s hi, lo;
    split4(v, hi, lo);
    *reinterpret_cast<v4h_s*>(up + (g.row0 + r) * KU + 4 * q) = hi;
    *reinterpret_cast<v4h_s*>(up + M * KU + (g.row0 + r) * KU + 4 * q) = lo;
  }
  return 0;
}

// once per sample: x planes of the stage input, zero pad columns of the u planes
__device__ __attribute__((noinline)) void xs_init_phase(XsArgs kp, int n_) {
  const int n = __builtin_amdgcn_readfirstlane(n_);
  const XsArgs p = xs_args(kp);
  const XsGeo g = xs_geo(p, n);
  const int tid = threadIdx.x;
  const int C = p->C, Ds = p->Ds, KU = p->KSC * 16;
  const long M = p->M;
  const float* xin = p->xin;
  _Float16* xp = p->xp;
  _Float16* up = p->up;
  for (int i = tid; i < g.R * (C >> 2); i += 256) {
    const int r = i / (C >> 2), c = (i - r * (C >> 2)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(xin + (g.row0 + r) * C + c);
    v4h_s hi, lo;
    split4(v, hi, lo);
    *reinterpret_cast<v4h_s*>(xp + (g.row0 + r) * C + c) = hi;
    *reinterpret_cast<v4h_s*>(xp + M * C + (g.row0 + r) * C + c) = lo;
  }
  if (KU > Ds) {
    const int padq = (KU - Ds) >> 2;
    for (int i = tid; i < g.R * padq; i += 256) {
      const int r = i / padq, c = Ds + (i - r * padq) * 4;
      const v4h_s z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      *reinterpret_cast<v4h_s*>(up + (g.row0 + r) * KU + c) = z;
      *reinterpret_cast<v4h_s*>(up + M * KU + (g.row0 + r) * KU + c) = z;
    }
  }
}

// MAXA / MAXC: output chunks (32 columns) one wave accumulates in phase A / C; RPI: rows of a depthwise item
template <int MAXA, int MAXC, int RPI>
__global__ __launch_bounds__(256, 2) void x3d_stage_kernel(const X3dStageArgs p) {
  const XsArgs kp = (XsArgs)__builtin_amdgcn_kernarg_segment_ptr();     // == &p
  int bad = 0;
  for (int n = blockIdx.x & 7; n < p.N; n += 8) {
    xs_init_phase(kp, n);
    int se_idx = 0;
#pragma unroll 1
    for (int k = 0; k < p.nblocks; ++k) {
      bad |= xs_gemm_phase<MAXA, 0>(kp, n, k, 0);
      if (xs_dw_phase<RPI>(kp, n, k)) goto aborted;
      if ((p.se_mask >> k) & 1u) {
        if (xs_se_phase(kp, n, k, se_idx)) goto aborted;
        ++se_idx;
      }
      bad |= xs_gemm_phase<MAXC, 1>(kp, n, k, k == 0 ? 1 : 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  report_nonfinite(p.status, bad != 0);
  return;
aborted:
  if (threadIdx.x == 0 && p.status) *reinterpret_cast<volatile int*>(p.status) = 2;     // workgroups were not co-resident: output invalid
}

}  // namespace mspi

using namespace mspi;

namespace {

struct StageGeom {
  int Ds, KSA, NA, KSC, NC, KPA, KPC, TH, tiles_f, P, RPI, ring, inst;
  size_t lds;
  bool ok;
};

StageGeom stage_geom(const MspiX3dStageDesc* d) {
  StageGeom g;
  memset(&g, 0, sizeof(g));
  if (!d || d->N < 1 || d->T < 1 || d->T > 32 || d->H < 1 || d->W < 1 || d->W > 14 || d->C < 32 || d->C % 32 || d->C > 256 ||
      d->D < 4 || d->D > 512 || d->F < 1 || d->F > 32 || d->nblocks < 1 || d->nblocks > 32)
    return g;
  g.Ds = (d->D + 3) / 4 * 4;
  g.KSA = d->C / 16; g.NA = (g.Ds + 31) / 32;
  g.KSC = (g.Ds + 15) / 16; g.NC = d->C / 32;
  int n = 32 / d->T;
  if (n > d->H) n = d->H;
  if (n < 1) return g;
  g.TH = (d->H + n - 1) / n;
  g.tiles_f = (d->H + g.TH - 1) / g.TH;
  g.P = d->T * g.tiles_f;
  const int R = g.TH * d->W;
  if (R > 128 || g.P > 32) return g;
  // LDS ring stage: k-steps per stage so that a stage stays within 28 KB (at least one k-step, at most 4)
  const int cap = 28 * 1024;
  g.KPA = cap / (g.NA * 2048); if (g.KPA > 4) g.KPA = 4;
  g.KPC = cap / (g.NC * 2048); if (g.KPC > 4) g.KPC = 4;
  if (g.KPA < 1 || g.KPC < 1) return g;
  g.ring = (g.KPA * g.NA > g.KPC * g.NC ? g.KPA * g.NA : g.KPC * g.NC) * 2048;
  const int RG = (R + 31) / 32, RGe = RG == 3 ? 4 : RG, WPR = 4 / RGe;
  const int ma = (g.NA + WPR - 1) / WPR, mc = (g.NC + WPR - 1) / WPR;
  // depthwise item = RPI rows x one strip (<= 7 columns) x 4 channels, one item per thread: 4 rows when that still keeps
  // most threads busy, else 2
  const int nst = (d->W + 6) / 7, NQ = g.Ds / 4;
  g.RPI = ((g.TH + 3) / 4) * nst * NQ >= 160 ? 4 : 2;
  if (((g.TH + g.RPI - 1) / g.RPI) * nst * NQ > 256) return g;
  if (ma <= 4 && mc <= 2) g.inst = 0;          // <4, 2>
  else if (ma <= 7 && mc <= 3) g.inst = 1;     // <7, 3>
  else return g;
  g.lds = (size_t)2 * g.ring + 256 * 16 + (size_t)(2 * g.Ds + d->F) * 4 + 16;
  g.ok = g.lds <= 64 * 1024;
  return g;
}

size_t al256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

extern "C" int mspi_x3d_stage_supported(const MspiX3dStageDesc* d) { return stage_geom(d).ok ? 1 : 0; }

extern "C" size_t mspi_x3d_stage_packed_bytes(const MspiX3dStageDesc* d, size_t* float_params_per_block) {
  const StageGeom g = stage_geom(d);
  if (!g.ok) return 0;
  if (float_params_per_block)
    *float_params_per_block = (size_t)g.NA * 32 + d->C + g.Ds + 27 * (size_t)g.Ds + (size_t)d->F * g.Ds + d->F + (size_t)g.Ds * d->F + g.Ds + 2;
  return ((size_t)g.KSA * g.NA + (size_t)g.KSC * g.NC) * 2048;
}

extern "C" size_t mspi_x3d_stage_ws_bytes(const MspiX3dStageDesc* d) {
  const StageGeom g = stage_geom(d);
  if (!g.ok) return 0;
  const size_t M = (size_t)d->N * d->T * d->H * d->W;
  return al256(((size_t)d->N * g.P + d->N + 1) * 4) + al256(2 * M * g.Ds * 4) + al256(M * g.Ds * 4) + al256(2 * M * g.KSC * 16 * 2) +
         al256(2 * M * d->C * 2) + al256((size_t)2 * d->N * g.P * g.Ds * 4);
}

extern "C" int mspi_x3d_stage_fwd(const MspiX3dStageDesc* d, const void* x, void* y, const void* wq, const void* wf, void* ws,
                                  void* stream) {
  MSPI_REQUIRE(d && x && y && wq && wf && ws, "mspi_x3d_stage_fwd: null argument");
  const StageGeom g = stage_geom(d);
  MSPI_REQUIRE(g.ok, "mspi_x3d_stage_fwd: shape N=%d T=%d H=%d W=%d C=%d D=%d F=%d blocks=%d is outside the stage kernel's range",
               d->N, d->T, d->H, d->W, d->C, d->D, d->F, d->nblocks);
  MSPI_REQUIRE(aligned16(x) && aligned16(y) && aligned16(wq) && aligned16(wf) && aligned16(ws) && x != y, "mspi_x3d_stage_fwd: 16-byte alignment; x and y distinct");
  const size_t M = (size_t)d->N * d->T * d->H * d->W;
  MSPI_REQUIRE(2 * M * g.Ds * 4 < (1ull << 31), "mspi_x3d_stage_fwd: tensor too large for 32-bit buffer offsets");
  X3dStageArgs a;
  memset(&a, 0, sizeof(a));
  unsigned char* w = (unsigned char*)ws;
  const size_t sync_bytes = al256(((size_t)d->N * g.P + d->N + 1) * 4);
  a.sync = (unsigned*)w; w += sync_bytes;
  a.tbuf = (float*)w; w += al256(2 * M * g.Ds * 4);
  a.ubuf = (float*)w; w += al256(M * g.Ds * 4);
  a.up = (_Float16*)w; w += al256(2 * M * g.KSC * 16 * 2);
  a.xp = (_Float16*)w; w += al256(2 * M * d->C * 2);
  a.pool = (float*)w;
  a.xin = (const float*)x; a.y = (float*)y; a.wq = (const unsigned char*)wq; a.wf = (const float*)wf;
  size_t fpb = 0;
  a.wq_stride = (long)mspi_x3d_stage_packed_bytes(d, &fpb);
  a.wf_stride = (long)((fpb + 3) / 4 * 4);
  a.M = (long)M;
  a.N = d->N; a.T = d->T; a.H = d->H; a.W = d->W; a.C = d->C; a.D = d->D; a.Ds = g.Ds; a.F = d->F;
  a.nblocks = d->nblocks; a.se_mask = d->se_mask;
  a.TH = g.TH; a.tiles_f = g.tiles_f; a.P = g.P;
  a.KSA = g.KSA; a.NA = g.NA; a.KSC = g.KSC; a.NC = g.NC; a.KPA = g.KPA; a.KPC = g.KPC; a.RPI = g.RPI; a.ring_bytes = g.ring;
  a.status = g_status_word;
  hipStream_t s = (hipStream_t)stream;
  // every polled word starts at zero on every call (a memset node under graph capture, replayed first)
  if (hipMemsetAsync(a.sync, 0, sync_bytes, s) != hipSuccess) {
    (void)hipGetLastError();
    set_error("mspi_x3d_stage_fwd: hipMemsetAsync failed");
    return MSPI_ELAUNCH;
  }
  const dim3 grid(8 * g.P), block(256);
#define X3DS_LAUNCH(MA, MC, RP)                                                                                             \
  do {                                                                                                                      \
    hipLaunchKernelGGL((x3d_stage_kernel<MA, MC, RP>), grid, block, g.lds, s, a);                                            \
  } while (0)
  if (g.inst == 0) { if (g.RPI == 4) X3DS_LAUNCH(4, 2, 4); else X3DS_LAUNCH(4, 2, 2); }
  else { if (g.RPI == 4) X3DS_LAUNCH(7, 3, 4); else X3DS_LAUNCH(7, 3, 2); }
#undef X3DS_LAUNCH
  return check_launch("mspi_x3d_stage_fwd");
}
