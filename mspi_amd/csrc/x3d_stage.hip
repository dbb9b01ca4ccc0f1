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
//       A  t = relu(a(x))            f16x3 MFMA (32x32x16): B operand = the rows' f16 hi/lo planes, A operand = weight
//                                     fragments, both global -> registers through a ring of prefetch stages (no LDS, no barrier)
//       B  u = b(t) (+ pool sums)    fp32 FMAs from an LDS image of the tile's t cells + the 1-position halo owned by the
//                                     neighbouring tiles, CQ channel quads at a time, the next image's loads in flight meanwhile
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
  unsigned* sync;         // [N*P] t epochs | [N] pool arrivals | abort word | [8][P] XCC ids | [8] arrivals (zeroed by the launch function)
  const unsigned char* wq; const float* wf;
  long wq_stride, wf_stride;
  long M;
  int N, T, H, W, C, D, Ds, F;
  int nblocks;
  unsigned se_mask;
  int TH, tiles_f, P;
  int KSA, NA, KSC, NC;
  int* status;
  int force_wt;           // MSPI_X3D_STAGE_WT=1: write-through hand-offs even when a sample's workgroups share an XCD (A/B, tests)
  unsigned long long* stamps;   // diagnostic (tools/x3d_stage_debug.py): [workgroup][block][16] s_memrealtime stamps (100 MHz); NULL in production
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

// plain buffer store: the bytes stay (dirty) in this XCD's L2 -- for hand-offs whose every reader sits on the same XCD
__device__ __forceinline__ void st_l2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
  const f32x4_s f = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_s, f), r, (int)byte_off, 0, 0);
}

__device__ __forceinline__ void split4(float4 v, v4h_s& hi, v4h_s& lo) {
  _Float16 h, l;
  split_f16(v.x, h, l); hi[0] = h; lo[0] = l;
  split_f16(v.y, h, l); hi[1] = h; lo[1] = l;
  split_f16(v.z, h, l); hi[2] = h; lo[2] = l;
  split_f16(v.w, h, l); hi[3] = h; lo[3] = l;
}
__device__ __forceinline__ float swish1(float v) { return v / (1.f + __expf(-v)); }

// The kernel is a loop over (sample, block) whose body is four phases.  Written naively as one function the optimiser
// unswitched the block loop three ways and hoisted ~380 uniform values (launch parameters and everything derived from them) out
// of it -- hundreds of spilled SGPRs and VGPRs; as real (non-inlined) functions each phase saved and restored ~110
// callee-saved VGPRs per call (~1.7 us of scratch traffic per call).  So the phases ARE inlined, but each one starts from an
// OPAQUE copy of the kernarg pointer (an empty asm redefines it): nothing derived from the parameters is loop-invariant
// any more, every phase re-reads what it needs with scalar loads and re-derives its geometry, and live ranges end with the
// phase.
typedef const __attribute__((address_space(4))) X3dStageArgs* XsArgs;
#define XS_PHASE static __device__ __forceinline__
__device__ __forceinline__ XsArgs xs_args(XsArgs kp) {
  asm volatile("" : "+s"(kp));
  return kp;
}

// phase stamps of the diagnostic build path (p->stamps != NULL): one lane per workgroup, a value that nothing else reads
__device__ __forceinline__ void xs_stamp(XsArgs p, int k, int idx) {
  unsigned long long* st = p->stamps;
  if (st && threadIdx.x == 0) st[((long)blockIdx.x * p->nblocks + k) * 16 + idx] = __builtin_amdgcn_s_memrealtime();
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

__device__ __forceinline__ unsigned char* xs_lds() {
  extern __shared__ __attribute__((aligned(16))) unsigned char xs_smem[];
  return xs_smem;
}

// ---------------------------------------------------------------------------------------------------------------------
// GEMM phase.  PHASE 0 (A): t = relu(a(x) + ba) -> tbuf[k & 1] (write-through), then the tile's epoch is published.
//              PHASE 1 (C): x = relu(c(u planes) + bc + x) -> y (fp32) and the x planes.
// Wave (rg, cw) computes the output chunks cw, cw + WPR, ... of row group rg, four chunks at a time.  Both operands go
// global -> registers: the B operand is the rows' hi / lo planes (the workgroup's own data, L1), the A operand the weight
// fragments [ks][chunk][hi,lo][lane][8] (read-only; the 32 workgroups of a sample stream the same bytes, so they come from
// L2).  No LDS and no barrier inside the phase: the k loop is fully unrolled (KS is a template parameter) over a ring of PF
// register stages, loads for k-step ks + PF are issued right behind the MFMAs of k-step ks and the compiler's counted
// vmcnt waits keep PF - 1 stages in flight.  (First version: a two-stage LDS ring filled by LDS-DMA with a barrier per
// stage -- one exposed memory latency per stage, 26 of them per block in stage 5: 73 us per block.)
// Returns 1 when a result was not finite.
struct XsGemmCtx {
  const unsigned char* wq;        // this lane's 16 B inside the fragments of k-step 0, chunk 0
  const _Float16 *bhi, *blo;      // this lane's 8 halves of k-step 0 in the hi / lo plane
  const float* bias;              // ba / bc
  const float* xcur;              // phase C: residual source
  float* y; _Float16* xp;         // phase C: outputs
  __amdgpu_buffer_rsrc_t rs_t;    // phase A: t of this block's parity
  long M, xrow;                   // phase C: first float of the lane's row in x / y
  unsigned trow;                  // phase A: byte offset of the lane's row in t
  float inv_s;
  int C, Ds, cw, WPR, lh;
  bool rvalid, wt;
};

// One pass of NCG output chunks (g0 .. g0 + NCG - 1 of the wave's list; nv of them are real, the others repeat the last
// chunk and are dropped).  Every load and MFMA of the pass is unconditional, so the compiler counts its vmcnt waits exactly and
// PF - 1 register stages stay in flight (with `if (i < nc)` around the loads it could not count them and waited vmcnt(0) before
// every k-step).  The k loop is ROLLED (PF k-steps per trip, the last trips peeled at compile time): fully unrolled, the two
// GEMM phases alone were 40 KB of straight-line code executed once per block -- the whole kernel 60-74 KB against a 64 KB
// instruction cache shared by two CUs, and every phase ran ~10 us behind its instruction fetch.
template <int KS, int NCH, int PHASE, int NCG>
__device__ __forceinline__ bool xs_gemm_pass(const XsGemmCtx& c, int g0, int nv) {
  constexpr int PF = 3;
  constexpr int NMAIN = (KS - PF) / PF;              // trips whose PF prefetches all exist
  long joff[NCG];                                    // byte offset of chunk i's fragments inside one k-step
#pragma unroll
  for (int i = 0; i < NCG; ++i) joff[i] = (long)(c.cw + c.WPR * (g0 + min(i, nv - 1))) * 2048;
  v16f_s acc[NCG];
#pragma unroll
  for (int i = 0; i < NCG; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  v8h_s wh[PF][NCG], wl[PF][NCG], fh[PF], fl[PF];
  const unsigned char* wk = c.wq;                    // fragments of the k-step the next load fetches
  const _Float16* bh = c.bhi;
  const _Float16* bl = c.blo;
#define XS_LOAD(slot_)                                                                          \
  do {                                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < NCG; ++i_) {                                        \
      wh[slot_][i_] = *reinterpret_cast<const v8h_s*>(wk + joff[i_]);                           \
      wl[slot_][i_] = *reinterpret_cast<const v8h_s*>(wk + joff[i_] + 1024);                    \
    }                                                                                           \
    fh[slot_] = *reinterpret_cast<const v8h_s*>(bh);                                            \
    fl[slot_] = *reinterpret_cast<const v8h_s*>(bl);                                            \
    wk += NCH * 2048; bh += 16; bl += 16;                                                       \
  } while (0)
#define XS_MFMA(slot_)                                                                          \
  do {                                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < NCG; ++i_) {                                        \
      if (!kSingleProduct) {                                                                    \
        acc[i_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[slot_][i_], fh[slot_], acc[i_], 0, 0, 0); \
        acc[i_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[slot_][i_], fl[slot_], acc[i_], 0, 0, 0); \
      }                                                                                         \
      acc[i_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[slot_][i_], fh[slot_], acc[i_], 0, 0, 0);   \
    }                                                                                           \
  } while (0)
  // sched_barrier: the machine scheduler otherwise sinks every load to just above its first use (less register pressure,
  // and the whole prefetch distance gone: one MFMA of cover per load)
#pragma unroll
  for (int s = 0; s < PF; ++s) XS_LOAD(s);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int t = 0; t < NMAIN; ++t) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      XS_MFMA(s);
      __builtin_amdgcn_sched_barrier(0);
      XS_LOAD(s);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int ks = NMAIN * PF; ks < KS; ++ks) {         // the last PF .. 2 PF - 1 k-steps
    XS_MFMA(ks % PF);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + PF < KS) XS_LOAD(ks % PF);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef XS_LOAD
#undef XS_MFMA
  // epilogue: lane (li, lh) holds, for row rg*32 + li, columns j*32 + q*8 + 4*lh + 0..3
  bool bad = false;
  if (PHASE == 0) {
#pragma unroll
    for (int i = 0; i < NCG; ++i) {
      if (i >= nv) break;
      const int j = c.cw + c.WPR * (g0 + i);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = j * 32 + q * 8 + 4 * c.lh;
        const float4 b = *reinterpret_cast<const float4*>(c.bias + col);
        float4 v;
        v.x = fmaf(acc[i][q * 4 + 0], c.inv_s, b.x); v.y = fmaf(acc[i][q * 4 + 1], c.inv_s, b.y);
        v.z = fmaf(acc[i][q * 4 + 2], c.inv_s, b.z); v.w = fmaf(acc[i][q * 4 + 3], c.inv_s, b.w);
        bad |= c.rvalid && (nonfinite(v.x) | nonfinite(v.y) | nonfinite(v.z) | nonfinite(v.w));
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (c.rvalid && col < c.Ds) {
          if (c.wt) st_sc1(c.rs_t, c.trow + (unsigned)col * 4u, v);
          else st_l2(c.rs_t, c.trow + (unsigned)col * 4u, v);
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NCG; ++i) {
      if (i >= nv) break;
      const int j = c.cw + c.WPR * (g0 + i);
      float4 rv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        rv[q] = c.rvalid ? *reinterpret_cast<const float4*>(c.xcur + c.xrow + j * 32 + q * 8 + 4 * c.lh) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = j * 32 + q * 8 + 4 * c.lh;
        const float4 b = *reinterpret_cast<const float4*>(c.bias + col);
        float4 v;
        v.x = fmaf(acc[i][q * 4 + 0], c.inv_s, b.x) + rv[q].x; v.y = fmaf(acc[i][q * 4 + 1], c.inv_s, b.y) + rv[q].y;
        v.z = fmaf(acc[i][q * 4 + 2], c.inv_s, b.z) + rv[q].z; v.w = fmaf(acc[i][q * 4 + 3], c.inv_s, b.w) + rv[q].w;
        bad |= c.rvalid && (nonfinite(v.x) | nonfinite(v.y) | nonfinite(v.z) | nonfinite(v.w));
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (c.rvalid) {
          *reinterpret_cast<float4*>(c.y + c.xrow + col) = v;
          v4h_s hi, lo;
          split4(v, hi, lo);
          *reinterpret_cast<v4h_s*>(c.xp + c.xrow + col) = hi;
          *reinterpret_cast<v4h_s*>(c.xp + c.M * c.C + c.xrow + col) = lo;
        }
      }
    }
  }
  return bad;
}

template <int KS, int NCH, int PHASE>
XS_PHASE int xs_gemm_phase(XsArgs kp, int n_, int k_, int first_, int wt_) {
  const int n = n_, k = k_;
  const bool first = first_ != 0;
  const XsArgs p = xs_args(kp);
  const XsGeo g = xs_geo(p, n);
  const XsParams f = xs_params(p, k);
  const int tid = threadIdx.x, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int LDP = KS * 16;                       // plane row pitch in halves: C (phase A), 16 KSC (phase C)
  const long M = p->M;
  const int brow = min(g.rg * 32 + li, g.R - 1);     // rows past the tile repeat its last row; their results are dropped
  XsGemmCtx c;
  c.wq = p->wq + (long)k * p->wq_stride + (PHASE == 0 ? 0L : (long)p->KSA * p->NA * 2048) + lane * 16;
  c.bhi = (PHASE == 0 ? p->xp : p->up) + (g.row0 + brow) * LDP + 8 * lh;
  c.blo = c.bhi + M * LDP;
  c.M = M; c.C = p->C; c.Ds = p->Ds; c.cw = g.cw; c.WPR = g.WPR; c.lh = lh;
  c.rvalid = g.rg * 32 + li < g.R;
  c.wt = wt_ != 0;
  if (PHASE == 0) {
    c.bias = f.ba; c.inv_s = f.b2[p->Ds];
    c.rs_t = __builtin_amdgcn_make_buffer_rsrc(p->tbuf + (long)(k & 1) * M * p->Ds, 0, (int)(M * p->Ds * 4), 0x00020000);
    c.trow = (unsigned)((g.row0 + g.rg * 32 + li) * p->Ds) * 4u;
    c.xcur = nullptr; c.y = nullptr; c.xp = nullptr; c.xrow = 0;
  } else {
    c.bias = f.bc; c.inv_s = f.b2[p->Ds + 1];
    c.xcur = first ? p->xin : p->y; c.y = p->y; c.xp = p->xp;
    c.xrow = (g.row0 + g.rg * 32 + li) * p->C;
    c.rs_t = __builtin_amdgcn_make_buffer_rsrc(p->y, 0, 0, 0x00020000); c.trow = 0;
  }
  bool bad = false;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // the planes this phase reads are complete (all waves' stores drained)
  if (PHASE == 0) xs_stamp(p, k, 0);
  const int mych = max(0, (NCH - g.cw + g.WPR - 1) / g.WPR);
#pragma unroll 1
  for (int g0 = 0; g0 < mych; g0 += 4) {
    const int nc = min(4, mych - g0);                // chunks of this pass (wave-uniform)
    if (nc > 2) bad |= xs_gemm_pass<KS, NCH, PHASE, 4>(c, g0, nc);
    else bad |= xs_gemm_pass<KS, NCH, PHASE, 2>(c, g0, nc);
  }
  if (PHASE == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // EVERY storing wave drains before the barrier ...
    __syncthreads();
    if (tid == 0)                                            // ... and ONE lane publishes the tile's epoch
      __hip_atomic_store(p->sync + (long)n * p->P + g.tile, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  xs_stamp(p, k, PHASE == 0 ? 1 : 5);
  return bad ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Channel configuration of a stage (compile time: the k loops are unrolled, and every LDS / global offset inside the depthwise
// phase's loops is an instruction immediate -- with run-time pitches the optimiser hoisted ~150 precomputed addresses out of
// the chunk loop and spilled them).  HRM / WCM: rows / columns of the LDS image of one frame (tile + halo), at most.
template <int VAR> struct XsCfg;
template <> struct XsCfg<0> {   // X3D-L stage 4
  static constexpr int C = 96, DS = 216, KSA = 6, NA = 7, KSC = 14, NC = 3, CQ = 8, CGW = 4, HRM = 9, WCM = 16;
};
template <> struct XsCfg<1> {   // X3D-L stage 5
  static constexpr int C = 192, DS = 432, KSA = 12, NA = 14, KSC = 27, NC = 6, CQ = 16, CGW = 2, HRM = 6, WCM = 9;
};

// Depthwise phase: waits for the neighbouring tiles' t, then u = b(t) + bb on the tile, CQ channel quads (4 CQ channels) at a
// time: the chunk's t cells with the 1-position halo -- 3 frames x (th + 2) rows x (W + 2) columns, zeros outside the clip --
// are brought into LDS by ONE batch of write-through-coherent loads per thread (up to 14 in flight; the first version read
// its window straight from L2, nine loads at a time: 18 exposed latencies per block), the next chunk's batch is issued before
// this chunk is computed.  Thread = (quad, row, group of CGW columns): the 3 x 3 x (CGW + 2) window slides through registers
// from LDS.  Blocks without squeeze-excite write swish(u) as planes; SE blocks write fp32 u and the tile's pool sums (one
// write-through row per tile, added in a fixed order).  Returns 0, or 1 on a timeout.
template <int VAR>
XS_PHASE int xs_dw_phase(XsArgs kp, int n_, int k_, int se_idx_) {
  typedef XsCfg<VAR> G;
  constexpr int CQ = G::CQ, CGW = G::CGW, Ds = G::DS, KU = G::KSC * 16, NQ = Ds / 4;
  constexpr int RP = G::WCM * CQ * 16 + 128;             // LDS row pitch: +128 B so that neighbouring rows start on the other bank half
  constexpr int NROW = 3 * G::HRM, NEL = NROW * G::WCM * CQ;
  constexpr int NL = (NEL + 255) / 256;                   // halo loads per thread and chunk
  constexpr int NWL = (27 * CQ + 255) / 256;              // weight loads per thread and chunk
  constexpr int NCHUNK = (NQ + CQ - 1) / CQ;
  const int n = n_, k = k_, se_idx = se_idx_;
  const XsArgs p = xs_args(kp);
  const XsGeo g = xs_geo(p, n);
  const XsParams f = xs_params(p, k);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = p->W, H = p->H, T = p->T;
  const long M = p->M;
  const bool se = (p->se_mask >> k) & 1u;
  unsigned char* halo = xs_lds();
  unsigned char* wlds = halo + NROW * RP + 256;          // [27][CQ] float4 (256 B slack: the last group's window may run past its row)
  float* red = reinterpret_cast<float*>(wlds + 27 * CQ * 16);      // [256] float4
  int* sflag = reinterpret_cast<int*>(red + 256 * 4);

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
  xs_stamp(p, k, 2);

  const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(p->tbuf + (long)(k & 1) * M * Ds, 0, (int)(M * Ds * 4), 0x00020000);
  float* prow = p->pool + (((long)(se_idx & 1) * p->N + n) * p->P) * Ds;
  const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(prow, 0, p->P * Ds * 4, 0x00020000);
  constexpr unsigned XS_OOB = 0x40000000u;               // past num_records (< 2^30 - 2^16, checked on the host): the load returns zeros
  const int qc = tid % CQ;

  // this thread's halo elements: e = tid + 256 i  ->  image row e / (WCM CQ) = (frame dt, row ir), column ic, quad qc
  unsigned goff[NL];                                      // byte offset of the cell in tbuf (quad 0), or out of range
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int e = tid + 256 * i;
    const int row = e / (G::WCM * CQ), ic = (e / CQ) % G::WCM;
    const int dt = row / G::HRM, ir = row % G::HRM;
    const int tt = g.tf + dt - 1, hh = g.h0 + ir - 1, ww = ic - 1;
    const bool ok = e < NEL && ir < g.th + 2 && tt >= 0 && tt < T && hh >= 0 && hh < H && ww >= 0 && ww < W;
    goff[i] = ok ? (unsigned)((((long)n * T + tt) * g.HW + (long)hh * W + ww) * (long)(Ds * 4)) : 2u * XS_OOB;
  }
  // this thread's item
  const int NCG = (W + CGW - 1) / CGW;
  const int rest = tid / CQ;
  const int r = rest % g.th, cg = rest / g.th;            // adjacent lane groups are adjacent rows (bank halves alternate)
  const bool item = cg < NCG;
  const int w0 = cg * CGW;
  const unsigned char* hbase = halo + r * RP + (w0 * CQ + qc) * 16;     // window origin: (frame 0, image row r, column w0)
  const unsigned char* wbase = wlds + qc * 16;
  const long orow = g.row0 + (long)r * W + w0;             // first output position of the item

  float4 pf[NL], pw[NWL];
#define XS_ISSUE(c_)                                                                                                    \
  do {                                                                                                                  \
    const unsigned qo_ = (unsigned)((c_) * CQ + qc) * 16u;                                                              \
    const bool qok_ = (c_) * CQ + qc < NQ;                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < NL; ++i_) pf[i_] = ld_sc1(rs_t, qok_ ? goff[i_] + qo_ : 2u * XS_OOB);       \
    _Pragma("unroll") for (int i_ = 0; i_ < NWL; ++i_) {                                                                \
      const int idx_ = tid + 256 * i_;                         /* (tap, quad of the chunk) */                           \
      const int tap_ = idx_ / CQ, qq_ = idx_ - tap_ * CQ;                                                               \
      const bool ok_ = idx_ < 27 * CQ && (c_) * CQ + qq_ < NQ;                                                          \
      pw[i_] = ok_ ? *reinterpret_cast<const float4*>(f.wb + tap_ * Ds + 4 * ((c_) * CQ + qq_)) : make_float4(0.f, 0.f, 0.f, 0.f); \
    }                                                                                                                   \
  } while (0)
  XS_ISSUE(0);
#pragma unroll 1
  for (int c = 0; c < NCHUNK; ++c) {
    if (c == 3) xs_stamp(p, k, 8);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = tid + 256 * i;
      if (e < NEL) *reinterpret_cast<float4*>(halo + (e / (G::WCM * CQ)) * RP + (e % (G::WCM * CQ)) * 16) = pf[i];
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i)
      if (tid + 256 * i < 27 * CQ) *reinterpret_cast<float4*>(wlds + (tid + 256 * i) * 16) = pw[i];
    if (c == 3) xs_stamp(p, k, 9);
    __syncthreads();
    if (c == 3) xs_stamp(p, k, 10);
    if (c + 1 < NCHUNK) XS_ISSUE(c + 1);
    __builtin_amdgcn_sched_barrier(0);                    // the next chunk's loads are issued HERE, not sunk below the compute
    if (c == 3) xs_stamp(p, k, 11);
    const int quad = c * CQ + qc;
    float4 ps = make_float4(0.f, 0.f, 0.f, 0.f);
    if (item && quad < NQ) {
      float4 acc[CGW];
      const float4 bq = *reinterpret_cast<const float4*>(f.bb + 4 * quad);
#pragma unroll
      for (int j = 0; j < CGW; ++j) acc[j] = bq;
#pragma unroll 1
      for (int dt = 0; dt < 3; ++dt)                  // rolled: code size (the instruction cache holds the whole kernel)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          float4 win[CGW + 2], wv[3];
#pragma unroll
          for (int cc = 0; cc < CGW + 2; ++cc) win[cc] = *reinterpret_cast<const float4*>(hbase + (dt * G::HRM + kh) * RP + cc * CQ * 16);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) wv[kw] = *reinterpret_cast<const float4*>(wbase + ((dt * 3 + kh) * 3 + kw) * CQ * 16);
#pragma unroll
          for (int j = 0; j < CGW; ++j)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              acc[j].x = fmaf(win[j + kw].x, wv[kw].x, acc[j].x);
              acc[j].y = fmaf(win[j + kw].y, wv[kw].y, acc[j].y);
              acc[j].z = fmaf(win[j + kw].z, wv[kw].z, acc[j].z);
              acc[j].w = fmaf(win[j + kw].w, wv[kw].w, acc[j].w);
            }
          __builtin_amdgcn_sched_barrier(0);          // one tap row's LDS reads in flight, not all nine rows' (324 registers)
        }
      if (se) {
        float* up_ = p->ubuf + orow * Ds + 4 * quad;
#pragma unroll
        for (int j = 0; j < CGW; ++j) {
          if (w0 + j < W) {
            const float4 v = acc[j];
            ps.x += v.x; ps.y += v.y; ps.z += v.z; ps.w += v.w;
            *reinterpret_cast<float4*>(up_ + j * Ds) = v;
          }
        }
      } else {
        _Float16* hp = p->up + orow * KU + 4 * quad;
        _Float16* lp = hp + M * KU;
#pragma unroll
        for (int j = 0; j < CGW; ++j) {
          if (w0 + j < W) {
            float4 v = acc[j];
            v.x = swish1(v.x); v.y = swish1(v.y); v.z = swish1(v.z); v.w = swish1(v.w);
            v4h_s hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<v4h_s*>(hp + j * KU) = hi;
            *reinterpret_cast<v4h_s*>(lp + j * KU) = lo;
          }
        }
      }
    }
    if (se) *reinterpret_cast<float4*>(red + tid * 4) = ps;
    if (c == 3) xs_stamp(p, k, 12);
    __syncthreads();                                      // everybody is done with this chunk's LDS image; `red` is complete
    if (c == 3) xs_stamp(p, k, 13);
    if (se && tid < CQ && c * CQ + tid < NQ) {            // the tile's pool sums of this chunk's quads, fixed order
      float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int m = 0; m < 256 / CQ; ++m) {
        const float4 v = *reinterpret_cast<const float4*>(red + (m * CQ + tid) * 4);
        s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
      }
      st_sc1(rs_p, (unsigned)(g.tile * Ds + 4 * (c * CQ + tid)) * 4u, s4);
    }
  }
#undef XS_ISSUE
  xs_stamp(p, k, 3);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Squeeze-excite phase of an SE block: arrive (the tile's pool row was written by the depthwise phase), wait for the sample's
// P rows, gate = sigmoid(fc2(relu(fc1(mean)))) (every workgroup adds the P rows in the same order), then u -> swish(g u) as
// planes.  Every loop issues its loads in batches (a load -> use -> load chain was 30 us of exposed latency here).
// Returns 0, or 1 on a timeout.
template <int VAR>
XS_PHASE int xs_se_phase(XsArgs kp, int n_, int k_, int se_idx_) {
  typedef XsCfg<VAR> G;
  constexpr int Ds = G::DS, NQ = Ds / 4, KU = G::KSC * 16;
  constexpr int QP = NQ <= 64 ? 64 : 128, PARTS = 256 / QP, RPP = 32 / PARTS;      // pool rows: PARTS threads per quad, RPP rows each
  const int n = n_, k = k_, se_idx = se_idx_;
  const XsArgs p = xs_args(kp);
  const XsGeo g = xs_geo(p, n);
  const XsParams f = xs_params(p, k);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int F = p->F, P = p->P;
  const long M = p->M;
  float* smean = reinterpret_cast<float*>(xs_lds());     // the depthwise phase's LDS image is dead by now
  float* sgate = smean + Ds;
  float* shid = sgate + Ds;
  float* spart = shid + 32;                               // [PARTS][Ds]
  int* sflag = reinterpret_cast<int*>(spart + PARTS * Ds);
  unsigned* pcount = p->sync + (long)p->N * P + n;
  float* prow = p->pool + (((long)(se_idx & 1) * p->N + n) * P) * Ds;
  const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(prow, 0, P * Ds * 4, 0x00020000);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the storing waves drain (pool row, u)
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(pcount, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (wave == 0) {
    const bool ok = wave_wait_ge(lane == 0 ? pcount : nullptr, (unsigned)(P * (se_idx + 1)), p->sync + (long)p->N * P + p->N);
    if (lane == 0) *sflag = ok ? 1 : 0;
  }
  __syncthreads();
  if (*sflag == 0) return 1;
  xs_stamp(p, k, 6);
  {   // mean over the sample: thread (quad, part) adds the rows part, part + PARTS, ... (one batch of loads), then the parts in order
    const int q = tid % QP, part = tid / QP;
    float4 v[RPP];
#pragma unroll
    for (int i = 0; i < RPP; ++i) {
      const int r = part + PARTS * i;
      v[i] = ld_sc1(rs_p, (q < NQ && r < P) ? (unsigned)(r * Ds + 4 * q) * 4u : 0x80000000u);      // out of range: zeros
    }
    float4 s4 = v[0];
#pragma unroll
    for (int i = 1; i < RPP; ++i) { s4.x += v[i].x; s4.y += v[i].y; s4.z += v[i].z; s4.w += v[i].w; }
    if (q < NQ) *reinterpret_cast<float4*>(spart + part * Ds + 4 * q) = s4;
  }
  __syncthreads();
  if (tid < NQ) {
    float4 s4 = *reinterpret_cast<const float4*>(spart + 4 * tid);
#pragma unroll
    for (int i = 1; i < PARTS; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(spart + i * Ds + 4 * tid);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    const float inv = 1.f / (float)((long)p->T * g.HW);
    s4.x *= inv; s4.y *= inv; s4.z *= inv; s4.w *= inv;
    *reinterpret_cast<float4*>(smean + 4 * tid) = s4;
  }
  __syncthreads();
  {   // fc1 + ReLU: F <= 32 hidden units, 8 lanes each, float4 loads of the weight row in one batch
    constexpr int NI = (NQ + 7) / 8;
    const int hf = tid >> 3, part = tid & 7;
    const int hfc = min(hf, F - 1);
    float4 w[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = min(part + 8 * i, NQ - 1);
      w[i] = *reinterpret_cast<const float4*>(f.w1 + (long)hfc * Ds + 4 * q);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = part + 8 * i;
      if (q < NQ) {
        const float4 m = *reinterpret_cast<const float4*>(smean + 4 * q);
        s = fmaf(w[i].x, m.x, s); s = fmaf(w[i].y, m.y, s); s = fmaf(w[i].z, m.z, s); s = fmaf(w[i].w, m.w, s);
      }
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (hf < F && part == 0) shid[hf] = fmaxf(s + f.b1[hf], 0.f);
  }
  __syncthreads();
  {   // fc2 + sigmoid: thread = channel (two passes when Ds > 256), the F weights of a channel are contiguous
#pragma unroll
    for (int c0 = 0; c0 < Ds; c0 += 256) {
      const int c = min(c0 + tid, Ds - 1);
      float4 w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = *reinterpret_cast<const float4*>(f.w2 + (long)c * F + 4 * min(i, F / 4 - 1));
      float s = f.b2[c];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (4 * i < F) {
          s = fmaf(w[i].x, shid[4 * i], s); s = fmaf(w[i].y, shid[4 * i + 1], s);
          s = fmaf(w[i].z, shid[4 * i + 2], s); s = fmaf(w[i].w, shid[4 * i + 3], s);
        }
      }
      if (c0 + tid < Ds) sgate[c] = 1.f / (1.f + __expf(-s));
    }
  }
  __syncthreads();
  const float* ubuf = p->ubuf + g.row0 * Ds;
  _Float16* up = p->up + g.row0 * KU;
  const int total = g.R * NQ;
#pragma unroll 1
  for (int i0 = tid; i0 < total; i0 += 256 * 6) {         // six independent loads per thread and pass
    float4 v[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int i = min(i0 + 256 * j, total - 1);
      v[j] = *reinterpret_cast<const float4*>(ubuf + (long)(i / NQ) * Ds + 4 * (i % NQ));
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int i = i0 + 256 * j;
      if (i < total) {
        const int r = i / NQ, q = i % NQ;
        const float4 gq = *reinterpret_cast<const float4*>(sgate + 4 * q);
        float4 u = v[j];
        u.x = swish1(u.x * gq.x); u.y = swish1(u.y * gq.y); u.z = swish1(u.z * gq.z); u.w = swish1(u.w * gq.w);
        v4h_s hi, lo;
        split4(u, hi, lo);
        *reinterpret_cast<v4h_s*>(up + (long)r * KU + 4 * q) = hi;
        *reinterpret_cast<v4h_s*>(up + M * KU + (long)r * KU + 4 * q) = lo;
      }
    }
  }
  __syncthreads();                                        // sgate is read to the end before the next phase reuses the LDS
  xs_stamp(p, k, 4);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Placement check, once per workgroup: do the workgroups of this group (one sample at a time) share an XCD?  If they do, the t
// tiles are handed over through that XCD's L2 (plain stores, `sc1` loads: nothing leaves the die -- with write-through stores
// every tile's halo, 4.4x the tensor, came back from the Infinity Cache: 95 MB per block).  If they do not, or if the check
// times out, the write-through form is used: the result never depends on where the workgroups run.  Returns 1 = write-through.
XS_PHASE int xs_placement(XsArgs kp) {
  const XsArgs p = xs_args(kp);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = p->P, grp = blockIdx.x & 7, tile = blockIdx.x >> 3;
  unsigned* tab = p->sync + (long)p->N * P + p->N + 1;    // [8][P] XCC id + 1, then [8] arrivals
  unsigned* cnt = tab + 8 * P;
  int* sflag = reinterpret_cast<int*>(xs_lds());
  const unsigned mine = (__builtin_amdgcn_s_getreg(6164) & 15u) + 1u;      // HW_REG_XCC_ID (id 20), bits 3:0
  if (tid == 0) {
    __hip_atomic_store(tab + grp * P + tile, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(cnt + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (wave == 0) {
    const bool ok = wave_wait_ge(lane == 0 ? cnt + grp : nullptr, (unsigned)P, p->sync + (long)p->N * P + p->N);
    const unsigned v = (ok && lane < P) ? ld_relaxed(tab + grp * P + lane) : mine;
    const bool same = ok && __all(v == mine);
    if (lane == 0) *sflag = same ? 0 : 1;
  }
  __syncthreads();
  const int wt = *sflag;
  __syncthreads();
  return wt;
}

// once per sample: x planes of the stage input, zero pad columns of the u planes
XS_PHASE void xs_init_phase(XsArgs kp, int n_) {
  const int n = n_;
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

// VAR 0: C = 96, dim_inner 216 (X3D-L stage 4);  VAR 1: C = 192, dim_inner 432 (stage 5)
template <int VAR>
__global__ __launch_bounds__(256, 2) void x3d_stage_kernel(const X3dStageArgs p) {
  typedef XsCfg<VAR> G;
  const XsArgs kp = (XsArgs)__builtin_amdgcn_kernarg_segment_ptr();     // == &p
  int bad = 0;
  if ((int)(blockIdx.x & 7) >= p.N) return;                             // fewer samples than XCD groups
  const int wt = p.force_wt ? 1 : xs_placement(kp);
  for (int n = blockIdx.x & 7; n < p.N; n += 8) {
    xs_init_phase(kp, n);
    int se_idx = 0;
#pragma unroll 1
    for (int k = 0; k < p.nblocks; ++k) {
      bad |= xs_gemm_phase<G::KSA, G::NA, 0>(kp, n, k, 0, wt);
      if (xs_dw_phase<VAR>(kp, n, k, se_idx)) goto aborted;
      if ((p.se_mask >> k) & 1u) {
        if (xs_se_phase<VAR>(kp, n, k, se_idx)) goto aborted;
        ++se_idx;
      }
      bad |= xs_gemm_phase<G::KSC, G::NC, 1>(kp, n, k, k == 0 ? 1 : 0, wt);
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
  int Ds, KSA, NA, KSC, NC, TH, tiles_f, P, var;
  size_t lds;
  bool ok;
};

StageGeom stage_geom(const MspiX3dStageDesc* d) {
  StageGeom g;
  memset(&g, 0, sizeof(g));
  if (!d || d->N < 1 || d->T < 1 || d->T > 32 || d->H < 1 || d->W < 1 || d->W > 14 || d->F < 1 || d->F > 32 || d->nblocks < 1 ||
      d->nblocks > 32)
    return g;
  // the k loops are unrolled at compile time: the two channel configurations of X3D-L's stride-1 stages with C % 32 == 0
  if (d->C == 96 && d->D == 216) g.var = 0;
  else if (d->C == 192 && d->D == 432) g.var = 1;
  else return g;
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
  const int CQ = g.var == 0 ? 8 : 16, CGW = g.var == 0 ? 4 : 2, HRM = g.var == 0 ? 9 : 6, WCM = g.var == 0 ? 16 : 9;
  if (g.TH + 2 > HRM || d->W + 2 > WCM) return g;                   // the LDS image of a frame (tile + halo) has a fixed pitch
  if (g.TH * ((d->W + CGW - 1) / CGW) * CQ > 256) return g;         // one depthwise item per thread
  const size_t halo = (size_t)3 * HRM * (WCM * CQ * 16 + 128) + 256;
  g.lds = halo + 27 * CQ * 16 + 256 * 16 + 16;
  const size_t se = (size_t)((2 + 4) * g.Ds + 32) * 4 + 16;
  if (se > halo) return g;
  g.ok = g.lds <= 80 * 1024;                                        // two such workgroups fit a CU side by side
  return g;
}

size_t al256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

// diagnostic: a device buffer of 8 * nblocks * grid uint64 for phase stamps (NULL switches them off again)
static unsigned long long* g_xs_stamps = nullptr;
extern "C" int mspi_x3d_stage_debug_stamps(void* buf) { g_xs_stamps = (unsigned long long*)buf; return MSPI_OK; }

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
  return al256(((size_t)d->N * g.P + d->N + 1 + 8 * g.P + 8) * 4) + al256(2 * M * g.Ds * 4) + al256(M * g.Ds * 4) + al256(2 * M * g.KSC * 16 * 2) +
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
  MSPI_REQUIRE(M * g.Ds * 4 < (1ull << 30) - (1ull << 16), "mspi_x3d_stage_fwd: tensor too large for the kernel's 32-bit buffer offsets");
  X3dStageArgs a;
  memset(&a, 0, sizeof(a));
  unsigned char* w = (unsigned char*)ws;
  const size_t sync_bytes = al256(((size_t)d->N * g.P + d->N + 1 + 8 * g.P + 8) * 4);
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
  a.KSA = g.KSA; a.NA = g.NA; a.KSC = g.KSC; a.NC = g.NC;
  a.status = g_status_word;
  a.stamps = g_xs_stamps;
  static const int force_wt = getenv("MSPI_X3D_STAGE_WT") ? atoi(getenv("MSPI_X3D_STAGE_WT")) : 0;
  a.force_wt = force_wt;
  hipStream_t s = (hipStream_t)stream;
  // every polled word starts at zero on every call (a memset node under graph capture, replayed first)
  if (hipMemsetAsync(a.sync, 0, sync_bytes, s) != hipSuccess) {
    (void)hipGetLastError();
    set_error("mspi_x3d_stage_fwd: hipMemsetAsync failed");
    return MSPI_ELAUNCH;
  }
  const dim3 grid(8 * g.P), block(256);
  static bool attr_set[2] = {false, false};
  if (!attr_set[g.var]) {     // more than 64 KB of dynamic LDS needs the opt-in (once per process; not a stream operation)
    const hipError_t e = g.var == 0 ? hipFuncSetAttribute((const void*)x3d_stage_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)
                                    : hipFuncSetAttribute((const void*)x3d_stage_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("mspi_x3d_stage_fwd: cannot reserve %zu bytes of LDS", g.lds);
      return MSPI_ELAUNCH;
    }
    attr_set[g.var] = true;
  }
  if (g.var == 0) hipLaunchKernelGGL((x3d_stage_kernel<0>), grid, block, g.lds, s, a);
  else hipLaunchKernelGGL((x3d_stage_kernel<1>), grid, block, g.lds, s, a);
  return check_launch("mspi_x3d_stage_fwd");
}
