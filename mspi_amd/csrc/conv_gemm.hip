// Implicit-GEMM convolution / linear on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain, 64 FLOP/clk/SIMD).
//
//   y[m, co] = act( sum_k A[m,k] * w[co,k] + bias[co] (+ res[m,co]) )
//
// A is never materialised: the k index is (tap, ci) with ci fastest and a row m is an
// output position (n,to,ho,wo); the loader gathers A straight from the strided input.
// Tiling is for 64-wide wavefronts: a 256-thread workgroup = 4 waves, each wave owns
// TM x TN 32x32 accumulator tiles (16 VGPRs each, column on the lane).  The K loop moves
// BK=32 floats per step through LDS rows padded to 36 floats, so that the ds_read_b128
// fragment reads (lane (i,h) takes k = 4h..4h+3 of row i: the MFMA K order is permuted
// identically for A and B, which a dot product does not care about) are conflict-free.
#include "conv_common.h"
#include <type_traits>
#include <stdlib.h>

namespace mspi {

constexpr int LDK = 36;  // padded LDS row (floats): 144 B keeps 16-B alignment, kills b128 conflicts
constexpr int LDH = 32;  // F16X3: LDS row (halves): 64 B = four 16-B slots, slot s of row r stored at s ^ ((r >> 2) & 3) -- the
                         // LDS-DMA kernels' swizzle (0.000 measured bank conflicts on their ds_read_b128).  Round 1-2 padded the rows
                         // to 80 B instead: conflict-free for the 16-B fragment reads, but the 8-B hi / lo staging writes of two
                         // neighbouring rows then overlapped in 4 of 32 banks (SQ_LDS_BANK_CONFLICT 0.29-0.36 of the LDS cycles).
                         // At a 64-B pitch a 16-lane group of 8-B writes is two whole rows = all 32 banks once.
__device__ __forceinline__ int swz16(int row, int slot) { return (slot ^ ((row >> 2) & 3)) << 3; }   // halves

enum { LOAD_V4 = 0, LOAD_S = 1 };

template <int BM, int BN, int WM, int WN, int LOADER, int PREC>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_gemm_kernel(const ConvArgs p) {
  constexpr int NT = WM * WN * 64;   // threads
  constexpr int RP = NT / 8;         // tile rows staged per pass (8 threads x float4 cover the 32-float k chunk)
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int AR = BM / RP;        // A rows staged per thread
  constexpr int BR = (BN + RP - 1) / RP;  // B rows staged per thread (F32)
  static_assert(BM % RP == 0 && TM >= 1 && TN >= 1, "tile/wave shape");

  // F32: A|B tiles of floats, rows LDK.  F16X3: A_hi|A_lo|B_hi|B_lo tiles of halves, rows LDH (same bytes/row+pad).
  // Two LDS stages (one barrier per K step, global prefetch two tiles ahead).
  constexpr int STAGE = PREC == PREC_F32 ? (BM + BN) * LDK : (BM + BN) * LDH;   // floats per stage
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int logical = xcd_logical_block(blockIdx.x, p.nblocks);
  const int tile_n = logical % p.tiles_n;
  const int tile_m = logical / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int kv = tid & 7;      // which float4 of the 32-float k chunk
  const int rbase = tid >> 3;  // 0..RP-1

  // ---- per-row gather state (fixed over the K loop) ----
  long a_off[AR];
  int a_t[AR], a_h[AR], a_w[AR], a_n[AR];
  bool a_ok[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = m0 + rbase + RP * i;
    a_ok[i] = m < p.M;
    if (!a_ok[i]) m = 0;
    if (p.dense_rows) {      // rows of a matrix: no (n, t, h, w) decomposition (three integer divisions per row otherwise)
      a_n[i] = p.gate ? m / p.rows_per_sample : 0;      // (the squeeze-excite gate is per sample)
      a_t[i] = 0; a_h[i] = 0; a_w[i] = 0;
      a_off[i] = (long)m * p.sW;
      continue;
    }
    int wo = m % p.Wo;
    int t1 = m / p.Wo;
    int ho = t1 % p.Ho;
    int t2 = t1 / p.Ho;
    int to = t2 % p.To;
    int n = t2 / p.To;
    a_n[i] = n;
    a_t[i] = to * p.strT - p.padT;
    a_h[i] = ho * p.strH - p.padH;
    a_w[i] = wo * p.strW - p.padW;
    a_off[i] = (long)n * p.sN + (long)a_t[i] * p.sT + (long)a_h[i] * p.sH + (long)a_w[i] * p.sW;
  }
  const float* b_ptr[BR];
  bool b_ok[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    int n = n0 + rbase + RP * i;
    b_ok[i] = n < p.Cout && rbase + RP * i < BN;
    b_ptr[i] = p.w + (long)(b_ok[i] ? n : 0) * p.ldw;
  }
  // F16X3 weights: [2][Cout][ldw] halves (hi plane, lo plane); a thread stages 16-B segments
  // (8 halves) seg = tid&3 of rows (tid>>2) + 64*i.
  constexpr int HB = (BN * 4 + NT - 1) / NT;  // 16-B segments per thread per plane
  constexpr int HR = NT / 4;                  // weight rows staged per pass
  const _Float16* wh = reinterpret_cast<const _Float16*>(p.w);
  const long wplane = (long)p.Cout * p.ldw;
  const int hseg = tid & 3, hrow = tid >> 2;

  // ---- k cursor of this thread's float4 (V4 loader): (tap, c) and tap -> (dt,dh,dw) ----
  const int nk = (int)((p.ldw + BK - 1) / BK);          // ldw >= K by contract; [K, ldw) is zero in w and masked in A
  const int it_begin = p.ws ? (int)((long)nk * blockIdx.y / p.ksplit) : 0;
  const int it_end = p.ws ? (int)((long)nk * (blockIdx.y + 1) / p.ksplit) : nk;
  int kc = kv * 4 + it_begin * BK, ktap = 0, kdt = 0, kdh = 0, kdw = 0;
  if (LOADER == LOAD_V4 && p.dense_rows && it_begin == 0) {
    // one tap, cursor inside the first k-step: no divisions (kc < 32 <= ... ; past C it points at the zero padding of the weights)
    if (kc >= p.C) { kc -= p.C; ktap = 1; }
  } else if (LOADER == LOAD_V4) {
    ktap = kc / p.C;
    kc -= ktap * p.C;
    int khw = p.kH * p.kW;
    kdt = ktap / khw;
    int r = ktap - kdt * khw;
    kdh = r / p.kW;
    kdw = r - kdh * p.kW;
  }
  const int ntaps = p.kT * p.kH * p.kW;

  float4 ra[AR], rb[BR], rg[AR];
  uint4 rbh[HB], rbl[HB];

  auto load_tiles = [&](int k0) {
    const int k = k0 + kv * 4;
    if (PREC == PREC_F32) {
      // B (weights): rows are zero padded to ldw (multiple of 4)
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        // unconditional load from a always-valid address + select: a branch around the load would make
        // hipcc wait vmcnt(0) per load and serialise the whole tile fetch
        const bool ok = b_ok[i] && k < p.ldw;
        const float4 v = *reinterpret_cast<const float4*>(ok ? b_ptr[i] + k : p.w);
        rb[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < HB; ++i) {
        const int r = hrow + HR * i, n = n0 + r;
        const bool ok = r < BN && n < p.Cout;   // ldw is a multiple of BK: no k guard
        const _Float16* q = wh + (ok ? (long)n * p.ldw + k0 + hseg * 8 : 0);
        const uint4 h = *reinterpret_cast<const uint4*>(q);
        const uint4 l = *reinterpret_cast<const uint4*>(q + wplane);
        rbh[i] = ok ? h : make_uint4(0u, 0u, 0u, 0u);
        rbl[i] = ok ? l : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    if (LOADER == LOAD_V4) {
      const bool kin = ktap < ntaps;
      const long koff = (long)kdt * p.sT + (long)kdh * p.sH + (long)kdw * p.sW + kc;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const bool inb = kin && a_ok[i] && (unsigned)(a_t[i] + kdt) < (unsigned)p.T &&
                         (unsigned)(a_h[i] + kdh) < (unsigned)p.H && (unsigned)(a_w[i] + kdw) < (unsigned)p.W;
        const float4 v = *reinterpret_cast<const float4*>(p.x + (inb ? a_off[i] + koff : 0));
        ra[i] = inb ? v : make_float4(0.f, 0.f, 0.f, 0.f);   // swish(0 * g) = 0: masked rows stay zero under the gate
      }
      if (p.gate) {   // wave-uniform; the gate values ride with the A tile and are applied in store_tiles: applying them here
                      // made every k-step wait for its own loads (a `c` conv with a gate cost 27.6 us against 18.9 without)
#pragma unroll
        for (int i = 0; i < AR; ++i) rg[i] = *reinterpret_cast<const float4*>(p.gate + (long)a_n[i] * p.C + (kin ? kc : 0));
      }
      // advance the cursor by one BK chunk
      kc += BK;
      while (kc >= p.C) {
        kc -= p.C;
        ++ktap;
        if (++kdw == p.kW) {
          kdw = 0;
          if (++kdh == p.kH) {
            kdh = 0;
            ++kdt;
          }
        }
      }
    } else {
      float va[AR][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ke = k + e;
        const int tap = ke / p.C;
        const int c = ke - tap * p.C;
        const int khw = p.kH * p.kW;
        const int dt = tap / khw;
        const int r = tap - dt * khw;
        const int dh = r / p.kW;
        const int dw = r - dh * p.kW;
        const bool kin = tap < ntaps;
        const long koff = (long)dt * p.sT + (long)dh * p.sH + (long)dw * p.sW + (long)c * p.sC;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
          const bool inb = kin && a_ok[i] && (unsigned)(a_t[i] + dt) < (unsigned)p.T &&
                           (unsigned)(a_h[i] + dh) < (unsigned)p.H && (unsigned)(a_w[i] + dw) < (unsigned)p.W;
          const float t = p.x[inb ? a_off[i] + koff : 0];
          va[i][e] = inb ? t : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) ra[i] = make_float4(va[i][0], va[i][1], va[i][2], va[i][3]);
    }
  };

  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31, lh = lane >> 5;

  // registers -> LDS stage `st` (F16X3: split every fp32 activation into hi + lo halves on the way)
  auto store_tiles = [&](int st) {
    float* base = smem + st * STAGE;
    if (LOADER == LOAD_V4 && p.gate) {      // u' = swish(u * gate) (squeeze-excite + Swish of the X3D block in front of `c`)
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        ra[i].x = fast_swish(ra[i].x * rg[i].x);
        ra[i].y = fast_swish(ra[i].y * rg[i].y);
        ra[i].z = fast_swish(ra[i].z * rg[i].z);
        ra[i].w = fast_swish(ra[i].w * rg[i].w);
      }
    }
    if (PREC == PREC_F32) {
      float* As = base;
      float* Bs = base + BM * LDK;
#pragma unroll
      for (int i = 0; i < AR; ++i)
        *reinterpret_cast<float4*>(&As[(rbase + RP * i) * LDK + kv * 4]) = ra[i];
#pragma unroll
      for (int i = 0; i < BR; ++i)
        if (rbase + RP * i < BN) *reinterpret_cast<float4*>(&Bs[(rbase + RP * i) * LDK + kv * 4]) = rb[i];
    } else {
      _Float16* Ah = reinterpret_cast<_Float16*>(base);
      _Float16* Al = Ah + BM * LDH;
      _Float16* Bh = Al + BM * LDH;
      _Float16* Bl = Bh + BN * LDH;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        v4h hi, lo;
        const float r4[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 h, l;
          split_f16(r4[e], h, l);
          hi[e] = h; lo[e] = l;
        }
        if (p.dbg & 1) lo = hi;
        const int r = rbase + RP * i;
        const int o = r * LDH + swz16(r, kv >> 1) + (kv & 1) * 4;
        *reinterpret_cast<v4h*>(&Ah[o]) = hi;
        *reinterpret_cast<v4h*>(&Al[o]) = lo;
      }
#pragma unroll
      for (int i = 0; i < HB; ++i) {
        const int r = hrow + HR * i;
        if (r < BN) {
          *reinterpret_cast<uint4*>(&Bh[r * LDH + swz16(r, hseg)]) = rbh[i];
          *reinterpret_cast<uint4*>(&Bl[r * LDH + swz16(r, hseg)]) = rbl[i];
        }
      }
    }
  };

  // MFMAs of one half (kh = 0/1) of the K step held in LDS stage `st`
  auto compute_half = [&](int st, int kh) {
    const float* base = smem + st * STAGE;
    if (PREC == PREC_F32) {
      const float* As = base;
      const float* Bs = base + BM * LDK;
#pragma unroll
      for (int kq = 0; kq < 2; ++kq) {
        const int kk = kh * 2 + kq;
        float4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[i] = *reinterpret_cast<const float4*>(&As[((wm * TM + i) * 32 + li) * LDK + kk * 8 + lh * 4]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[j] = *reinterpret_cast<const float4*>(&Bs[((wn * TN + j) * 32 + li) * LDK + kk * 8 + lh * 4]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
          }
      }
    } else {
      // 32x32x16 f16: lane (i = lane&31, h = lane>>5) holds A[i][8h..8h+7] / B[8h..8h+7][j] of the 16-deep step
      const _Float16* Ah = reinterpret_cast<const _Float16*>(base);
      const _Float16* Al = Ah + BM * LDH;
      const _Float16* Bh = Al + BM * LDH;
      const _Float16* Bl = Bh + BN * LDH;
      v8h ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + li;
        const int o = r * LDH + swz16(r, kh * 2 + lh);
        ah[i] = *reinterpret_cast<const v8h*>(&Ah[o]);
        al[i] = *reinterpret_cast<const v8h*>(&Al[o]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = (wn * TN + j) * 32 + li;
        const int o = r * LDH + swz16(r, kh * 2 + lh);
        bh[j] = *reinterpret_cast<const v8h*>(&Bh[o]);
        bl[j] = *reinterpret_cast<const v8h*>(&Bl[o]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (!kSingleProduct) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  // Software pipeline: while stage `cur` is being multiplied, tile it+1 (already in registers) is split and
  // written to the other stage and tile it+2 is requested from memory -- one barrier per K step, two tiles of
  // global-load latency cover.
  if (it_begin < it_end) {
    load_tiles(it_begin * BK);
    store_tiles(0);
    if (it_begin + 1 < it_end) load_tiles((it_begin + 1) * BK);
    __syncthreads();
  }
  for (int it = it_begin; it < it_end; ++it) {
    const int cur = (it - it_begin) & 1;
    if (!(p.dbg & 8)) compute_half(cur, 0);
    if (it + 1 < it_end && !(p.dbg & 4)) store_tiles(cur ^ 1);
    if (it + 2 < it_end && !(p.dbg & 2)) load_tiles((it + 2) * BK);
    if (!(p.dbg & 8)) compute_half(cur, 1);
    __syncthreads();
  }
  if (p.ws) {   // split-K: raw partial sums, reduced in a fixed order by splitk_reduce_kernel
    float* wz = p.ws + (long)blockIdx.y * p.M * p.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
      if (col >= p.Cout) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int rb0 = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rb0 + (r & 3) + 8 * (r >> 2);
          if (row < p.M) wz[(long)row * p.Cout + col] = PREC == PREC_F32 ? acc[i][j][r] : acc[i][j][r] * p.out_scale;
        }
      }
    }
    return;
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  // the activation is dispatched once, outside the element loops (see conv_gemm_ad.hip)
  bool bad = false;
  auto epilogue = [&](auto act_c) {
  constexpr int ACT = decltype(act_c)::value;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + li;
    if (col >= p.Cout) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rb0 = m0 + (wm * TM + i) * 32 + 4 * lh;
      float rv[16];
      if (p.res) {   // all 16 residual loads in flight together (one wait), not load-wait-store per element
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rb0 + (r & 3) + 8 * (r >> 2);
          rv[r] = p.res[row < p.M ? (long)row * p.ldr + col : 0];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb0 + (r & 3) + 8 * (r >> 2);
        const float v = (PREC == PREC_F32 ? acc[i][j][r] : acc[i][j][r] * p.out_scale) + bv + rv[r];
        bad |= row < p.M && nonfinite(v);          // padding rows of the last tile carry no result: never flagged
        if (row < p.M) p.y[(long)row * p.ldy + col] = act_apply(v, ACT);
      }
    }
  }
  };
  switch (p.act) {
    case MSPI_ACT_RELU: epilogue(std::integral_constant<int, MSPI_ACT_RELU>()); break;
    case MSPI_ACT_GELU: epilogue(std::integral_constant<int, MSPI_ACT_GELU>()); break;
    case MSPI_ACT_SIGMOID: epilogue(std::integral_constant<int, MSPI_ACT_SIGMOID>()); break;
    case MSPI_ACT_SWISH: epilogue(std::integral_constant<int, MSPI_ACT_SWISH>()); break;
    default: epilogue(std::integral_constant<int, MSPI_ACT_NONE>()); break;
  }
  report_nonfinite(p.status, bad);
}

template <int BM, int BN, int WM, int WN>
static void launch_cfg(const ConvArgs& a, bool v4, int prec, hipStream_t s) {
  const dim3 g(a.nblocks, a.ws ? a.ksplit : 1), b(WM * WN * 64);
  if (prec == PREC_F32) {
    if (v4) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, LOAD_V4, PREC_F32>), g, b, 0, s, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, LOAD_S, PREC_F32>), g, b, 0, s, a);
  } else {
    if (v4) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, LOAD_V4, PREC_F16X3>), g, b, 0, s, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, LOAD_S, PREC_F16X3>), g, b, 0, s, a);
  }
}

}  // namespace mspi

using namespace mspi;

namespace mspi {
int launch_conv_ad(ConvArgs& a, long Ml, int force_bn, int* cfg, hipStream_t s);
int launch_conv_ad8(ConvArgs& a, long Ml, int bn, int* cfg, hipStream_t s);
int launch_conv_sp(ConvArgs& a, long Ml, int bn, int rows, int* cfg, hipStream_t s);
}

static thread_local int g_last_cfg = 0;
extern "C" int mspi_conv_last_config(void) { return g_last_cfg; }

namespace mspi {
// y = act( sum_z ws[z] + bias + res ): the K slices' partial sums, added in slice order (bitwise reproducible)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, long M, int Cout,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            long ldr, float* __restrict__ y, long ldy, int act, int* status) {
  const long total = M * (Cout >> 2);
  bool bad = false;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long row = idx / (Cout >> 2);
    const int c = (int)(idx - row * (Cout >> 2)) * 4;
    float4 a = *reinterpret_cast<const float4*>(ws + row * Cout + c);
    for (int z = 1; z < S; ++z) {
      const float4 b = *reinterpret_cast<const float4*>(ws + ((long)z * M + row) * Cout + c);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + c); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    if (res) { const float4 b = *reinterpret_cast<const float4*>(res + row * ldr + c); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    bad |= nonfinite(a.x) | nonfinite(a.y) | nonfinite(a.z) | nonfinite(a.w);
    if (act != MSPI_ACT_NONE) {
      auto fin = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
        a.x = act_apply(a.x, ACT); a.y = act_apply(a.y, ACT); a.z = act_apply(a.z, ACT); a.w = act_apply(a.w, ACT);
      };
      MSPI_DISPATCH_ACT(act, fin)
    }
    *reinterpret_cast<float4*>(y + row * ldy + c) = a;
  }
  report_nonfinite(status, bad);
}
}  // namespace mspi

static int conv_fwd_impl(const MspiConvDesc* d, const float* x, const float* w, const float* bias, const float* res,
                         const float* gate, float* y, float* ws, int ksplit, mspi_stream_t stream) {
  MSPI_REQUIRE(d && x && w && y, "mspi_conv_fwd: null argument");
  MSPI_REQUIRE(d->N > 0 && d->T > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->Cout > 0, "mspi_conv_fwd: empty extent");
  MSPI_REQUIRE(d->kT > 0 && d->kH > 0 && d->kW > 0 && d->strT > 0 && d->strH > 0 && d->strW > 0 && d->padT >= 0 &&
                   d->padH >= 0 && d->padW >= 0,
               "mspi_conv_fwd: bad kernel/stride/pad");
  const int To = (d->T + 2 * d->padT - d->kT) / d->strT + 1;
  const int Ho = (d->H + 2 * d->padH - d->kH) / d->strH + 1;
  const int Wo = (d->W + 2 * d->padW - d->kW) / d->strW + 1;
  MSPI_REQUIRE(To == d->To && Ho == d->Ho && Wo == d->Wo && To > 0 && Ho > 0 && Wo > 0,
               "mspi_conv_fwd: output extent (%d,%d,%d) does not match formula (%d,%d,%d)", d->To, d->Ho, d->Wo, To,
               Ho, Wo);
  const long K = (long)d->kT * d->kH * d->kW * d->C;
  MSPI_REQUIRE(d->prec == PREC_F32 || d->prec == PREC_F16X3, "mspi_conv_fwd: unknown precision mode %d", d->prec);
  MSPI_REQUIRE(d->ldw >= K && (d->ldw & 3) == 0 && aligned16(w), "mspi_conv_fwd: weight rows must be 16-B aligned, ldw >= K");
  MSPI_REQUIRE(d->prec == PREC_F32 || ((d->ldw % BK) == 0 && d->w_scale > 0.f),
               "mspi_conv_fwd: f16x3 weights need ldw %% 32 == 0 and a positive w_scale");
  MSPI_REQUIRE(d->ldy >= d->Cout, "mspi_conv_fwd: ldy < Cout");
  MSPI_REQUIRE(!res || d->ldr >= d->Cout, "mspi_conv_fwd: ldr < Cout");
  const long Ml = (long)d->N * To * Ho * Wo;
  MSPI_REQUIRE(Ml < (1L << 31) && K < (1L << 31), "mspi_conv_fwd: problem too large for 32-bit row index");

  const bool v4 = d->sC == 1 && (d->C & 3) == 0 && aligned16(x) && (d->sN & 3) == 0 && (d->sT & 3) == 0 &&
                  (d->sH & 3) == 0 && (d->sW & 3) == 0 && (!gate || aligned16(gate));
  MSPI_REQUIRE(!gate || (v4 && d->kT == 1 && d->kH == 1 && d->kW == 1),
               "mspi_conv_fwd: gate needs a 1x1x1 conv on a 16-B aligned channels-last input");

  ConvArgs a;
  a.x = x; a.w = w; a.bias = bias; a.res = res; a.gate = gate; a.y = y;
  a.N = d->N; a.T = d->T; a.H = d->H; a.W = d->W; a.C = d->C;
  a.sN = d->sN; a.sT = d->sT; a.sH = d->sH; a.sW = d->sW; a.sC = d->sC;
  a.kT = d->kT; a.kH = d->kH; a.kW = d->kW;
  a.strT = d->strT; a.strH = d->strH; a.strW = d->strW;
  a.padT = d->padT; a.padH = d->padH; a.padW = d->padW;
  a.To = To; a.Ho = Ho; a.Wo = Wo; a.Cout = d->Cout;
  a.ldy = d->ldy; a.ldw = d->ldw; a.ldr = d->ldr; a.act = d->act;
  a.M = (int)Ml; a.K = (int)K; a.rows_per_sample = To * Ho * Wo;
  a.out_scale = d->prec == PREC_F16X3 ? 1.0f / d->w_scale : 1.0f;
  a.status = g_status_word;
  static const int dbg = getenv("MSPI_CONV_DBG") ? atoi(getenv("MSPI_CONV_DBG")) : 0;
  a.dbg = dbg;
  a.ws = ws; a.ksplit = ksplit;
  a.dense_rows = (d->kT == 1 && d->kH == 1 && d->kW == 1 && d->strT == 1 && d->strH == 1 && d->strW == 1 && d->padT == 0 && d->padH == 0 &&
                  d->padW == 0 && d->sC == 1 && d->sH == (int64_t)d->W * d->sW && d->sT == (int64_t)d->H * d->sH &&
                  d->sN == (int64_t)d->T * d->sT) ? 1 : 0;
  a.wb = (d->prec == PREC_F16X3) ? (const _Float16*)d->w_blocked : nullptr; a.xs = nullptr; a.ldxs = 0; a.xplane = 0; a.ys = nullptr; a.ldys = 0; a.yplane = 0;
  if (ws) {
    // split-K (mspi_conv_splitk_fwd): 64x64 tiles, gridDim.y = ksplit slices of the K loop, then the ordered reduction
    MSPI_REQUIRE(!gate && (d->Cout & 3) == 0 && (d->ldy & 3) == 0 && (!res || (d->ldr & 3) == 0) && aligned16(y) && aligned16(ws) &&
                     (!res || aligned16(res)) && (!bias || aligned16(bias)),
                 "mspi_conv_splitk_fwd: no gate; Cout / ldy / ldr multiples of 4; 16-B aligned pointers");
    MSPI_REQUIRE(ksplit >= 2 && ksplit <= 64 && ksplit <= (d->ldw + BK - 1) / BK, "mspi_conv_splitk_fwd: ksplit = %d", ksplit);
    a.tiles_n = (d->Cout + 63) / 64;
    const long nb = ((Ml + 63) / 64) * a.tiles_n;
    MSPI_REQUIRE(nb < (1L << 31), "mspi_conv_splitk_fwd: grid too large");
    a.nblocks = (int)nb;
    hipStream_t s = (hipStream_t)stream;
    g_last_cfg = (64 << 16) | (64 << 4) | (d->prec << 1) | (v4 ? 0 : 1);
    launch_cfg<64, 64, 2, 2>(a, v4, d->prec, s);
    const long total = Ml * (d->Cout >> 2);
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(mspi::splitk_reduce_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, ws, ksplit,
                       Ml, d->Cout, bias, res, (long)d->ldr, y, (long)d->ldy, d->act, mspi::g_status_word);
    return check_launch("mspi_conv_splitk_fwd");
  }

  // LDS-DMA form (conv_gemm_ad.hip): measured faster than the register-staged kernel on deep implicit GEMMs
  // (multi-tap convs, K >= 2048: 208 vs 199 TFLOP/s on the 3x3x3 readout conv), slower on the 1x1x1 layers
  // (its per-stage address block costs more than it saves there) -- tools/gemm_probe.py.
  static const int dma_mode = getenv("MSPI_CONV_DMA") ? atoi(getenv("MSPI_CONV_DMA")) : 1;   // 0 never, 1 auto, 2 always
  const bool deep_conv = (long)d->kT * d->kH * d->kW > 1 && K >= 2048 && Ml >= 16384;
  const bool dma_ok = d->prec == PREC_F16X3 && v4;
  MSPI_REQUIRE(d->tile >= -1 && d->tile <= 14 && (d->tile < 6 || dma_ok) && (d->tile != 8 || d->Cout <= 256),
               "mspi_conv_fwd: tile %d not available for this call", d->tile);
  if (d->tile >= 12) {   // LDS-DMA kernel with a 256-row tile and 8 waves sharing one weight tile
    static const int bn8[3] = {256, 192, 128};
    int cfg = 0;
    const int rc = launch_conv_ad8(a, Ml, bn8[d->tile - 12], &cfg, (hipStream_t)stream);
    MSPI_REQUIRE(rc == 0, "mspi_conv_fwd: tile %d could not be launched", d->tile);
    g_last_cfg = cfg;
    return check_launch("mspi_conv_fwd");
  }
  if (dma_ok && (d->tile >= 6 || (d->tile < 0 && (dma_mode == 2 || (dma_mode == 1 && deep_conv))))) {
    static const int dma_bn[6] = {128, 64, 1, 96, 192, 32};   // tile 6..11 (1 = all columns in one tile)
    int cfg = 0;
    const int rc = launch_conv_ad(a, Ml, d->tile >= 6 ? dma_bn[d->tile - 6] : 0, &cfg, (hipStream_t)stream);
    if (rc >= 0) {
      g_last_cfg = cfg;
      return rc == 0 ? check_launch("mspi_conv_fwd") : rc;
    }
  }

  // Tile choice.  time ~ rounds x (work of one workgroup): rounds = ceil(blocks / resident slots) -- whole rounds,
  // because a 588-block grid on 512 slots takes as long as 1024 blocks would (measured: tools/gemm_probe.py) --
  // and per-workgroup work ~ bm*bn*(K + K0) with K0 standing for the prologue + epilogue.  All tiles sustain about
  // the same rate on big grids (eff), so the choice is mostly about padding waste and round quantisation.
  struct Cfg { int bm, bn, slots; float eff; };
  static const Cfg cfgs[6] = {{128, 128, 512, 1.00f}, {128, 64, 512, 1.06f}, {128, 32, 768, 1.30f}, {64, 64, 1024, 1.25f},
                              {128, 128, 512, 1.00f}, {256, 128, 256, 1.10f}};
  static const int force = getenv("MSPI_CONV_TILE") ? atoi(getenv("MSPI_CONV_TILE")) : -1;
  int best = 3;
  double best_cost = 1e300;
  for (int i = 0; i < 4; ++i) {
    const long tm = (Ml + cfgs[i].bm - 1) / cfgs[i].bm, tn = (d->Cout + cfgs[i].bn - 1) / cfgs[i].bn;
    const long blocks = tm * tn;
    double rounds = (double)blocks / cfgs[i].slots;
    if (rounds < 6.0) rounds = (double)((blocks + cfgs[i].slots - 1) / cfgs[i].slots);
    const double cost = rounds * cfgs[i].bm * cfgs[i].bn * cfgs[i].eff * ((double)K + 192.0);
    if (cost < best_cost) { best_cost = cost; best = i; }
  }
  if (best == 0 && !getenv("MSPI_CONV_4WAVE")) best = 4;   // 128x128 runs best with 8 waves (32x64 per wave, 4 waves/SIMD)
  if (force >= 0 && force < 6) best = force;
  if (d->tile >= 0) best = d->tile;
  const int BMs = cfgs[best].bm, BNs = cfgs[best].bn;
  a.tiles_n = (d->Cout + BNs - 1) / BNs;
  const long nb = ((Ml + BMs - 1) / BMs) * a.tiles_n;
  MSPI_REQUIRE(nb < (1L << 31), "mspi_conv_fwd: grid too large");
  a.nblocks = (int)nb;
  hipStream_t s = (hipStream_t)stream;
  g_last_cfg = (BMs << 16) | (BNs << 4) | (best >= 4 ? 8 : 0) | (d->prec << 1) | (v4 ? 0 : 1);
  switch (best) {
    case 0: launch_cfg<128, 128, 2, 2>(a, v4, d->prec, s); break;
    case 1: launch_cfg<128, 64, 2, 2>(a, v4, d->prec, s); break;
    case 2: launch_cfg<128, 32, 4, 1>(a, v4, d->prec, s); break;
    case 4: launch_cfg<128, 128, 4, 2>(a, v4, d->prec, s); break;   // 8 waves, 32x64 per wave
    case 5: launch_cfg<256, 128, 4, 2>(a, v4, d->prec, s); break;   // 8 waves, 64x64 per wave
    default: launch_cfg<64, 64, 2, 2>(a, v4, d->prec, s); break;
  }
  return check_launch("mspi_conv_fwd");
}

extern "C" int mspi_conv_fwd(const MspiConvDesc* d, const float* x, const float* w, const float* bias,
                             const float* res, const float* gate, float* y, mspi_stream_t stream) {
  return conv_fwd_impl(d, x, w, bias, res, gate, y, nullptr, 1, stream);
}

extern "C" size_t mspi_conv_splitk_ws_bytes(const MspiConvDesc* d, int32_t ksplit) {
  if (!d || ksplit < 2) return 0;
  return (size_t)ksplit * (size_t)d->N * d->To * d->Ho * d->Wo * d->Cout * sizeof(float);
}

extern "C" int mspi_conv_splitk_fwd(const MspiConvDesc* d, const float* x, const float* w, const float* bias,
                                    const float* res, float* y, void* workspace, int32_t ksplit, mspi_stream_t stream) {
  MSPI_REQUIRE(workspace, "mspi_conv_splitk_fwd: null workspace");
  return conv_fwd_impl(d, x, w, bias, res, nullptr, y, (float*)workspace, ksplit, stream);
}


// ---- pre-split activations -------------------------------------------------------------------------------------------
namespace mspi {
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long ldx, _Float16* __restrict__ out,
                                                          long ldo, long plane, long M, int K4) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * K4) return;
  const long m = idx / K4;
  const int c = (int)(idx - m * K4) * 4;
  const float4 v = *reinterpret_cast<const float4*>(x + m * ldx + c);
  const float a[4] = {v.x, v.y, v.z, v.w};
  v4h h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) { _Float16 hh, ll; split_f16(a[e], hh, ll); h[e] = hh; l[e] = ll; }
  const long o = plane_off(m, c, K4 >> 3);
  *reinterpret_cast<v4h*>(out + o) = h;
  *reinterpret_cast<v4h*>(out + plane + o) = l;
}
}  // namespace mspi

extern "C" int mspi_split_planes_fwd(const float* x, int64_t ldx, int64_t M, int32_t K, void* planes, int64_t ldo, int64_t plane,
                                     mspi_stream_t stream) {
  MSPI_REQUIRE(x && planes && M > 0 && K > 0 && (K & 31) == 0 && (ldx & 3) == 0 && ldo == K && plane >= (M + 15) / 16 * 16 * ldo &&
                   aligned16(x) && aligned16(planes) && (plane & 7) == 0,
               "mspi_split_planes_fwd: K a multiple of 32, ldo == K (blocked planes), 16-B aligned pointers, plane >= roundup16(M)*K");
  const long total = M * (K / 4);
  MSPI_REQUIRE((total + 255) / 256 < (1L << 31), "mspi_split_planes_fwd: grid too large");
  hipLaunchKernelGGL(mspi::split_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)ldx, (_Float16*)planes, (long)ldo, (long)plane, (long)M, K / 4);
  return check_launch("mspi_split_planes_fwd");
}

namespace mspi {
// the inverse hand-over: fp32 rows = hi + lo (22 bits of the value the producer split)
__global__ __launch_bounds__(256) void join_planes_kernel(const _Float16* __restrict__ in, long ldi, long plane, float* __restrict__ y,
                                                         long ldy, long M, int K4) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * K4) return;
  const long m = idx / K4;
  const int c = (int)(idx - m * K4) * 4;
  const long o = plane_off(m, c, K4 >> 3);
  const v4h h = *reinterpret_cast<const v4h*>(in + o);
  const v4h l = *reinterpret_cast<const v4h*>(in + plane + o);
  float4 v;
  v.x = (float)h[0] + (float)l[0]; v.y = (float)h[1] + (float)l[1]; v.z = (float)h[2] + (float)l[2]; v.w = (float)h[3] + (float)l[3];
  *reinterpret_cast<float4*>(y + m * ldy + c) = v;
}
}  // namespace mspi

extern "C" int mspi_join_planes_fwd(const void* planes, int64_t ldi, int64_t plane, int64_t M, int32_t K, float* y, int64_t ldy,
                                    mspi_stream_t stream) {
  MSPI_REQUIRE(planes && y && M > 0 && K > 0 && (K & 31) == 0 && ldi == K && (ldy & 3) == 0 && ldy >= K &&
                   plane >= (M + 15) / 16 * 16 * ldi && aligned16(planes) && aligned16(y) && (plane & 7) == 0,
               "mspi_join_planes_fwd: K a multiple of 32, ldi == K (blocked planes), 16-B aligned pointers, plane >= roundup16(M)*K");
  const long total = M * (K / 4);
  MSPI_REQUIRE((total + 255) / 256 < (1L << 31), "mspi_join_planes_fwd: grid too large");
  hipLaunchKernelGGL(mspi::join_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)planes, (long)ldi, (long)plane, y, (long)ldy, (long)M, K / 4);
  return check_launch("mspi_join_planes_fwd");
}

extern "C" int mspi_gemm_sp_fwd(const MspiConvDesc* d, const void* x_planes, int64_t ldx, int64_t xplane, const float* w,
                                const float* bias, const float* res, float* y, void* y_planes, int64_t ldys, int64_t yplane,
                                mspi_stream_t stream) {
  MSPI_REQUIRE(d && x_planes && w && (y || y_planes), "mspi_gemm_sp_fwd: null argument");
  MSPI_REQUIRE(d->kT == 1 && d->kH == 1 && d->kW == 1 && d->strT == 1 && d->strH == 1 && d->strW == 1 && d->padT == 0 &&
                   d->padH == 0 && d->padW == 0, "mspi_gemm_sp_fwd: a plain GEMM on rows (1x1x1, stride 1, no padding)");
  MSPI_REQUIRE(d->prec == PREC_F16X3 && d->w_scale > 0.f && d->C > 0 && (d->C % BK) == 0 && d->ldw == d->C,
               "mspi_gemm_sp_fwd: f16x3 weights, K a multiple of 32 with ldw == K");
  MSPI_REQUIRE(ldx == d->C && (xplane & 7) == 0 && aligned16(x_planes) && aligned16(w),
               "mspi_gemm_sp_fwd: blocked input planes have ldx == K");
  const long Ml = (long)d->N * d->T * d->H * d->W;
  const long Mp = (Ml + 15) / 16 * 16;
  MSPI_REQUIRE(Ml > 0 && Ml < (1L << 31) && xplane >= Mp * ldx, "mspi_gemm_sp_fwd: bad extent (plane >= roundup16(M) * K)");
  MSPI_REQUIRE(!y_planes || (ldys == d->Cout && yplane >= Mp * ldys && (d->Cout % 32) == 0 && (yplane & 7) == 0 &&
                              aligned16(y_planes)), "mspi_gemm_sp_fwd: blocked output planes need Cout %% 32 == 0, ldys == Cout, plane >= roundup16(M) * Cout");
  MSPI_REQUIRE(!y || d->ldy >= d->Cout, "mspi_gemm_sp_fwd: ldy < Cout");
  MSPI_REQUIRE(!res || d->ldr >= d->Cout, "mspi_gemm_sp_fwd: ldr < Cout");
  ConvArgs a;
  a.x = nullptr; a.w = w; a.bias = bias; a.res = res; a.gate = nullptr; a.y = y;
  a.N = d->N; a.T = d->T; a.H = d->H; a.W = d->W; a.C = d->C;
  a.sN = a.sT = a.sH = a.sW = 0; a.sC = 1;
  a.kT = a.kH = a.kW = 1; a.strT = a.strH = a.strW = 1; a.padT = a.padH = a.padW = 0;
  a.To = d->T; a.Ho = d->H; a.Wo = d->W; a.Cout = d->Cout;
  a.ldy = d->ldy; a.ldw = d->ldw; a.ldr = d->ldr; a.act = d->act;
  a.M = (int)Ml; a.K = d->C; a.rows_per_sample = d->T * d->H * d->W;
  a.out_scale = 1.0f / d->w_scale;
  a.status = g_status_word;
  a.dbg = 0; a.ws = nullptr; a.ksplit = 1;
  a.dense_rows = 1;
  a.wb = nullptr;      // (the pre-split form takes blocked weights through `w`)
  a.xs = (const _Float16*)x_planes; a.ldxs = ldx; a.xplane = xplane;
  a.ys = (_Float16*)y_planes; a.ldys = ldys; a.yplane = yplane;
  int bn, rows = 128;
  switch (d->tile) {
    case 6: bn = 128; break;
    case 7: bn = 64; break;
    case 9: bn = 96; break;
    case 10: bn = 192; break;
    case 11: bn = 256; break;
    case 12: bn = 256; rows = 256; break;
    case 13: bn = 192; rows = 256; break;
    case 14: bn = 128; rows = 256; break;
    default: bn = d->Cout <= 64 ? 64 : (d->Cout % 192 == 0 ? 192 : 128); break;
  }
  int cfg = 0;
  const int rc = launch_conv_sp(a, Ml, bn, rows, &cfg, (hipStream_t)stream);
  MSPI_REQUIRE(rc == 0, "mspi_gemm_sp_fwd: tile %d could not be launched", d->tile);
  g_last_cfg = cfg;
  return check_launch("mspi_gemm_sp_fwd");
}
