// f16x3 implicit GEMM, LDS-DMA form -- the production GEMM of the f16x3 path.
//
// Ablation of the register-staged kernel in conv_gemm.hip (tools/gemm_probe.py + MSPI_CONV_DBG) showed the LDS
// pipe, not the matrix pipe, at its limit: splitting the activations on the way IN and staging both operands
// costs 32 KB of ds_write (~80 B/clk) + 64 KB of ds_read per 128x128x32 step, plus ~200 VALU instructions, against
// 768 MFMA cycles.  This kernel
//   * moves both tiles HBM/L2 -> LDS with global_load_lds_dwordx4 (LDS-DMA): no VGPR staging, no ds_write, no
//     VALU; the implicit-GEMM gather is the per-lane SOURCE address, padding taps read a zero page; the LDS image
//     is kept conflict-free by an XOR swizzle applied on the source side (rule: linear destination, swizzled
//     source, same swizzle on the read);
//   * stages the activations as RAW fp32 and splits them into f16 hi/lo AFTER the fragment read, in registers:
//     a wave owns 32 output rows x all BN columns, so every activation is split exactly once per workgroup;
//   * K order inside a 32-deep stage is permuted so that lane (i, h) needs A[i][16h .. 16h+15]: MFMA step s takes
//     k = 16h + 8s .. +7 for both operands (a dot product does not care);
//   * double buffered: the DMA of stage it+1 flies under the MFMAs of stage it, one barrier per stage.
#include "conv_common.h"
#include <type_traits>
#include <stdlib.h>

namespace mspi {

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

typedef __attribute__((address_space(3))) void lds_void;

// DENSE: 1x1x1, stride 1, no padding -- a plain GEMM on rows; the per-stage source address is base + k, no tap cursor
// NW: waves per workgroup = 32-row slabs of the tile (4: 128 x BN, two workgroups per CU; 8: 256 x BN, one workgroup per
// CU whose eight waves share ONE weight tile -- 1.67x fewer staged bytes per MFMA and a 1.33x longer compute phase to
// cover the latency of the next stage's DMA)
template <int BN, bool GATE, bool DENSE, int NW, bool APRE = false>
__global__ __launch_bounds__(64 * NW, 2) void conv_gemm_dma_kernel(const ConvArgs p) {
  static_assert(!APRE || (DENSE && !GATE), "pre-split activations: plain GEMM on rows only");
  constexpr int BM = 32 * NW;
  constexpr int TN = BN / 32;
  constexpr int A_BYTES = BM * 32 * 4;          // raw fp32 activations: 128 rows x 32 k
  constexpr int P_BYTES = BN * 32 * 2;          // one f16 weight plane: BN rows x 32 k
  constexpr int STAGE = A_BYTES + 2 * P_BYTES;
  constexpr int HBI = (BN + 16 * NW - 1) / (16 * NW);   // weight DMA instructions per wave per plane (16 rows each)
  static_assert(BN % 32 == 0 && BN >= 32 && BN <= 256, "BN: multiple of 32, <= 256");
  // Ring depth: 3 stages (two K steps of DMA in flight behind a COUNTED vmcnt + raw s_barrier, so the barrier does
  // not drain the newest stage) where 3 stages still leave two workgroups per CU, else 2 stages.
#ifndef MSPI_DMA_NST_SP
#define MSPI_DMA_NST_SP 2
#endif
#ifndef MSPI_SP_READS_FIRST
#define MSPI_SP_READS_FIRST 0   // A/B builds: 1 = all fragment reads of a stage before its MFMAs, 2 = + MFMAs pinned before the DMA wait.
#endif                          // Same-box result on six layers (tile 7): 146/128/140/144/162/199 us vs 150/130/144/138/167/204 vs
                                // 149/126/140/146/160/202 -- the loop's instruction order is not what bounds this kernel
  // measured: a 3-deep ring (BN <= 64) loses a resident workgroup to LDS and is 15-25 % slower (fp32-A form, round 1);
  // -DMSPI_DMA_NST_SP=3 builds the pre-split 128 x 64 form with a 3-deep ring for an A/B (tools/sp_probe.py, MSPI_LIB_PATH)
  constexpr int NST = (APRE && BN == 64 && NW == 4) ? MSPI_DMA_NST_SP : 2;
  constexpr int DMA_PER_STAGE = 4 + 2 * HBI;    // upper bound of this thread's DMA instructions per stage
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int logical = xcd_logical_block(blockIdx.x, p.nblocks);
  const int tile_n = logical % p.tiles_n, tile_m = logical / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;   // BM = 32 NW
  const int ntaps = p.kT * p.kH * p.kW;

  // ---- activation DMA assignment: instruction j fills LDS chunks (j*4+wave)*64 + lane (16 B each):
  //      row r_j = (j*4+wave)*8 + lane/8, slot lane%8; the chunk stored in that slot is c = slot ^ ((r>>1)&7),
  //      which is the same for all four j -> ONE k cursor per thread.
  long a_off[4];
  int a_t[4], a_h[4], a_w[4];
  bool a_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int m = m0 + (j * NW + wave) * 8 + (lane >> 3);
    a_ok[j] = m < p.M;
    if (!a_ok[j]) m = 0;
    if (p.dense_rows) { a_t[j] = 0; a_h[j] = 0; a_w[j] = 0; a_off[j] = (long)m * p.sW; continue; }   // rows of a matrix: no integer divisions
    const int wo = m % p.Wo;
    const int t1 = m / p.Wo;
    const int ho = t1 % p.Ho;
    const int t2 = t1 / p.Ho;
    const int to = t2 % p.To;
    const int n = t2 / p.To;
    a_t[j] = to * p.strT - p.padT;
    a_h[j] = ho * p.strH - p.padH;
    a_w[j] = wo * p.strW - p.padW;
    a_off[j] = (long)n * p.sN + (long)a_t[j] * p.sT + (long)a_h[j] * p.sH + (long)a_w[j] * p.sW;
  }
  const int a_chunk = (lane & 7) ^ ((((wave * 8 + (lane >> 3)) >> 1)) & 7);
  int kc = a_chunk * 4, ktap = 0, kdt = 0, kdh = 0, kdw = 0;
  while (kc >= p.C) {
    kc -= p.C;
    ++ktap;
    if (++kdw == p.kW) { kdw = 0; if (++kdh == p.kH) { kdh = 0; ++kdt; } }
  }
  // ---- weight DMA assignment: plane chunks (i*4+wave)*64 + lane: row r = q/4, slot q%4, segment = slot ^ ((r>>2)&3)
  const _Float16* wh = reinterpret_cast<const _Float16*>(p.w);
  // blocked weights (the pre-split form always; the others when the caller supplies them): rows padded to 16
  const bool wblk = APRE || p.wb != nullptr;
  if (wblk && !APRE) wh = p.wb;
  const long wplane = wblk ? (long)((p.Cout + 15) / 16 * 16) * p.ldw : (long)p.Cout * p.ldw;
  const int b_seg = (lane & 3) ^ ((lane >> 4) & 3);

  auto issue_stage = [&](int st, int k0) {
    unsigned char* base = smem + st * STAGE;
    if (APRE) {
      // the A tile as two f16 planes of BM rows x 64 B, staged exactly like a weight plane: group q = 16 rows,
      // row r = 16 q + lane/4, slot lane%4 holds segment slot ^ ((r>>2)&3)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = i * NW + wave;
        const bool ok = m0 + q * 16 < p.M;       // blocked planes: the whole 16-row group exists (rows are allocated to a multiple of 16)
        const _Float16* src = p.xs + ((long)((m0 >> 4) + (ok ? q : 0)) * (p.ldxs >> 5) + (k0 >> 5)) * 512 + (lane >> 2) * 32 + b_seg * 8;
        const void* s_hi = ok ? (const void*)src : (const void*)g_zero16;
        const void* s_lo = ok ? (const void*)(src + p.xplane) : (const void*)g_zero16;
        __builtin_amdgcn_global_load_lds(s_hi, (lds_void*)(base + q * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(s_lo, (lds_void*)(base + A_BYTES / 2 + q * 1024), 16, 0, 0);
      }
    } else if (DENSE) {
      const int kk = k0 + a_chunk * 4;
      const bool kin = kk < p.C;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* src = (kin && a_ok[j]) ? p.x + a_off[j] + kk : g_zero16;
        __builtin_amdgcn_global_load_lds(src, (lds_void*)(base + (j * NW + wave) * 1024), 16, 0, 0);
      }
    } else {
      const bool kin = ktap < ntaps;
      const long koff = (long)kdt * p.sT + (long)kdh * p.sH + (long)kdw * p.sW + kc;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool inb = kin && a_ok[j] && (unsigned)(a_t[j] + kdt) < (unsigned)p.T &&
                         (unsigned)(a_h[j] + kdh) < (unsigned)p.H && (unsigned)(a_w[j] + kdw) < (unsigned)p.W;
        const float* src = inb ? p.x + a_off[j] + koff : g_zero16;
        __builtin_amdgcn_global_load_lds(src, (lds_void*)(base + (j * NW + wave) * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < HBI; ++i) {
      if ((i * NW + wave) * 16 < BN) {   // wave-uniform: the last group of 16 rows may not exist for this wave
        const int r = (i * NW + wave) * 16 + (lane >> 2);
        const int n = n0 + r;
        const bool ok = wblk ? n0 + (i * NW + wave) * 16 < p.Cout : n < p.Cout;
        const _Float16* q = wblk ? wh + ((long)((n0 >> 4) + (ok ? i * NW + wave : 0)) * (p.ldw >> 5) + (k0 >> 5)) * 512 + (lane >> 2) * 32 + b_seg * 8
                                 : wh + (long)(ok ? n : 0) * p.ldw + k0 + b_seg * 8;
        const void* s_hi = ok ? (const void*)q : (const void*)g_zero16;
        const void* s_lo = ok ? (const void*)(q + wplane) : (const void*)g_zero16;
        __builtin_amdgcn_global_load_lds(s_hi, (lds_void*)(base + A_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(s_lo, (lds_void*)(base + A_BYTES + P_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
      }
    }
    if (!DENSE) {
      kc += BK;
      while (kc >= p.C) {
        kc -= p.C;
        ++ktap;
        if (++kdw == p.kW) { kdw = 0; if (++kdh == p.kH) { kdh = 0; ++kdt; } }
      }
    }
  };

  v16f acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // this lane's fragment row and (for the squeeze-excite gate, 1x1x1 convs only) its sample
  const int frow = wave * 32 + li;
  const int fsw = (frow >> 1) & 7;
  int gm = m0 + frow;
  if (gm >= p.M) gm = p.M - 1;
  const float* gate_row = p.gate ? p.gate + (long)(gm / p.rows_per_sample) * p.C : nullptr;

  // ---- a 16-deep MFMA step `sub` of stage st covers k = 16*lh + 8*sub .. +7 of the stage.  It is split in three
  // pieces so that the fragment reads + f16 split of step s+1 can be interleaved with the MFMAs of step s (a wave
  // issues in order: LDS latency and the cvt/sub chain would otherwise sit exposed in front of every MFMA group).
  struct Frag {
    float4 v0, v1, g0, g1;
    v8h ah, al, bh[TN], bl[TN];
  };
  auto frag_read = [&](Frag& f, int st, int sub, int k0) {
    const unsigned char* base = smem + st * STAGE;
    if (APRE) {
      const _Float16* Ah = reinterpret_cast<const _Float16*>(base);
      const _Float16* Al = reinterpret_cast<const _Float16*>(base + A_BYTES / 2);
      const int oa = frow * 32 + ((((2 * lh + sub)) ^ ((frow >> 2) & 3)) << 3);
      f.ah = *reinterpret_cast<const v8h*>(&Ah[oa]);
      f.al = *reinterpret_cast<const v8h*>(&Al[oa]);
    } else {
      const float* As = reinterpret_cast<const float*>(base);
      const int c0 = 4 * lh + 2 * sub;   // 16-B chunk index of this lane's first 4 floats
      f.v0 = *reinterpret_cast<const float4*>(&As[frow * 32 + ((c0 ^ fsw) << 2)]);
      f.v1 = *reinterpret_cast<const float4*>(&As[frow * 32 + (((c0 + 1) ^ fsw) << 2)]);
    }
    const _Float16* Bh = reinterpret_cast<const _Float16*>(base + A_BYTES);
    const _Float16* Bl = reinterpret_cast<const _Float16*>(base + A_BYTES + P_BYTES);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int r = j * 32 + li;
      const int o = r * 32 + ((((2 * lh + sub)) ^ ((r >> 2) & 3)) << 3);
      f.bh[j] = *reinterpret_cast<const v8h*>(&Bh[o]);
      f.bl[j] = *reinterpret_cast<const v8h*>(&Bl[o]);
    }
    if (GATE) {
      const int kk = k0 + 16 * lh + 8 * sub;
      const bool ok = kk < p.C;
      f.g0 = *reinterpret_cast<const float4*>(gate_row + (ok ? kk : 0));
      f.g1 = *reinterpret_cast<const float4*>(gate_row + (ok ? kk + 4 : 0));
    }
  };
  auto frag_split = [&](Frag& f) {
    if (APRE) return;
    float a8[8] = {f.v0.x, f.v0.y, f.v0.z, f.v0.w, f.v1.x, f.v1.y, f.v1.z, f.v1.w};
    if (GATE) {
      const float g8[8] = {f.g0.x, f.g0.y, f.g0.z, f.g0.w, f.g1.x, f.g1.y, f.g1.z, f.g1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) a8[e] = act_apply(a8[e] * g8[e], MSPI_ACT_SWISH);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 h, l;
      split_f16(a8[e], h, l);
      f.ah[e] = h; f.al[e] = l;
    }
  };
  auto frag_mfma = [&](const Frag& f) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (!kSingleProduct) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al, f.bh[j], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah, f.bl[j], acc[j], 0, 0, 0);
      }
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah, f.bh[j], acc[j], 0, 0, 0);
    }
  };

  const int nk = (int)(p.ldw / BK);   // ldw is a multiple of BK; [K, ldw) is zero in w and reads the zero page in A
  Frag f0, f1;
  if (NST == 2) {
    issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
      const int cur = it & 1;
      if (it + 1 < nk) issue_stage(cur ^ 1, (it + 1) * BK);   // DMA of the next stage flies under this stage's MFMAs
      frag_read(f0, cur, 0, it * BK);
      frag_split(f0);
      // sub-step 1's reads + split interleaved with sub-step 0's MFMAs: per MFMA one LDS read and a few VALU
      frag_read(f1, cur, 1, it * BK);
      frag_split(f1);
      frag_mfma(f0);
      if (APRE && MSPI_SP_READS_FIRST) {
        // pre-split operands: there is no VALU to hide, only LDS latency.  Interleaved 1:1 the compiler put an
        // `s_waitcnt lgkmcnt(0)` in front of every MFMA (each fragment read's latency exposed behind ONE 32-cycle MFMA); with
        // all 4 + 8 TN/2... reads of the stage issued first, the MFMAs wait with counted lgkmcnt and only the first pays
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + 4 * TN, 0);   // every DS read of both sub-steps
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * TN, 0);       // then the MFMAs
      } else {
#pragma unroll
        for (int g = 0; g < 3 * TN; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
          if (!APRE) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // 5 VALU (the hi/lo split)
        }
      }
      frag_mfma(f1);
      if (APRE && MSPI_SP_READS_FIRST == 2) __builtin_amdgcn_sched_barrier(0);   // MFMAs are issued BEFORE the wait for the next stage's DMA
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // my DMAs have landed ...
      __syncthreads();                                         // ... and so have everybody else's; stage `cur` is free
    }
  } else {
    // 3-deep ring.  Every thread issues the same number of DMA instructions per stage (masked lanes read the zero
    // page, missing weight groups are wave-uniform skips accounted for below), so "all but the newest stage have
    // landed" is s_waitcnt vmcnt(<instructions of one stage>).  Raw s_barrier: __syncthreads() would add vmcnt(0).
    const int my_dma = 4 + 2 * ((wave * 16 < BN ? 1 : 0) + ((4 + wave) * 16 < BN ? 1 : 0) + ((8 + wave) * 16 < BN ? 1 : 0) +
                                ((12 + wave) * 16 < BN ? 1 : 0));
    issue_stage(0, 0);
    if (nk > 1) issue_stage(1, BK);
    if (nk > 1) {
      if (my_dma == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
      const int nxt2 = cur >= 1 ? cur - 1 : 2;   // (cur + 2) % 3
      if (it + 2 < nk) issue_stage(nxt2, (it + 2) * BK);
      frag_read(f0, cur, 0, it * BK);
      frag_split(f0);
      frag_read(f1, cur, 1, it * BK);
      frag_split(f1);
      frag_mfma(f0);
#pragma unroll
      for (int g = 0; g < 3 * TN; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      }
      frag_mfma(f1);
      // stage it+1 must have landed before anybody reads it; stage it+2 (just issued) may stay in flight
      if (it + 2 < nk) {
        if (my_dma == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur == 2 ? 0 : cur + 1;
    }
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  // The activation is dispatched ONCE, outside the element loops: with `p.act` looked at per element the compiler kept the
  // whole switch (scalar compares and branches, the sigmoid's IEEE division) around each of the 16 * TN values.
  bool bad = false;
  auto epilogue = [&](auto act_c) {
  constexpr int ACT = decltype(act_c)::value;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + j * 32 + li;
    if (col >= p.Cout) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
    const int rb0 = m0 + wave * 32 + 4 * lh;
    float rv[16];
    if (p.res) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb0 + (r & 3) + 8 * (r >> 2);
        rv[r] = p.res[row < p.M ? (long)row * p.ldr + col : 0];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = 0.f;
    }
    if (p.ys) {
      // split-plane output: lanes c and c+1 trade halves so that every store is one 4-B pair of neighbouring columns
      // (an f16 per lane would be twice the store instructions of the fp32 epilogue for half the bytes)
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const bool odd = li & 1;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        unsigned own[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int r = 2 * q + e;
          const float pre = acc[j][r] * p.out_scale + bv + rv[r];
          bad |= rb0 + (r & 3) + 8 * (r >> 2) < p.M && nonfinite(pre);      // padding rows are never flagged
          const float v = act_apply(pre, ACT);
          _Float16 h, l;
          split_f16(v, h, l);
          h2 pr = {h, l};
          own[e] = __builtin_bit_cast(unsigned, pr);
        }
        // even lane keeps row 2q and receives the partner's row 2q; odd lane keeps row 2q+1 and receives the partner's
        const unsigned got = (unsigned)__shfl_xor((int)(odd ? own[0] : own[1]), 1, 64);
        const unsigned mine = odd ? own[1] : own[0];
        const h2 a = __builtin_bit_cast(h2, odd ? got : mine), b = __builtin_bit_cast(h2, odd ? mine : got);   // columns c0, c0+1
        const int r = 2 * q + (odd ? 1 : 0);
        const int row = rb0 + (r & 3) + 8 * (r >> 2);
        const int c0 = col & ~1;
        if (row < p.M) {      // Cout is a multiple of 32 for plane outputs (checked on the host): both columns exist
          h2 hi = {a[0], b[0]}, lo = {a[1], b[1]};
          const long po = plane_off(row, c0, (int)(p.ldys >> 5));
          *reinterpret_cast<h2*>(p.ys + po) = hi;
          *reinterpret_cast<h2*>(p.ys + p.yplane + po) = lo;
        }
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rb0 + (r & 3) + 8 * (r >> 2);
      const float pre = acc[j][r] * p.out_scale + bv + rv[r];
      bad |= row < p.M && nonfinite(pre);
      if (row < p.M) p.y[(long)row * p.ldy + col] = act_apply(pre, ACT);
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

// returns 0 when launched, -100 when this path does not apply (caller falls back to conv_gemm.hip).
// force_bn: 0 = heuristic; else the column tile (a multiple of 32 up to 256; 1 = "all columns in one tile").
template <int BN, int NW = 4>
static void launch_bn(const ConvArgs& a, hipStream_t s) {
  const dim3 g(a.nblocks), b(64 * NW);
  const bool dense = a.kT * a.kH * a.kW == 1 && a.strT == 1 && a.strH == 1 && a.strW == 1 && a.padT == 0 && a.padH == 0 &&
                     a.padW == 0;
  if (a.gate) hipLaunchKernelGGL((conv_gemm_dma_kernel<BN, true, true, NW>), g, b, 0, s, a);   // the gate implies 1x1x1
  else if (dense) hipLaunchKernelGGL((conv_gemm_dma_kernel<BN, false, true, NW>), g, b, 0, s, a);
  else hipLaunchKernelGGL((conv_gemm_dma_kernel<BN, false, false, NW>), g, b, 0, s, a);
}

// pre-split activations (mspi_gemm_sp_fwd): dense GEMM on rows, A = two f16 planes
template <int BN, int NW>
static void launch_sp(const ConvArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((conv_gemm_dma_kernel<BN, false, true, NW, true>), dim3(a.nblocks), dim3(64 * NW), 0, s, a);
}

int launch_conv_sp(ConvArgs& a, long Ml, int bn, int rows, int* cfg, hipStream_t s) {
  a.tiles_n = (int)((a.Cout + bn - 1) / bn);
  const long nb = ((Ml + rows - 1) / rows) * a.tiles_n;
  if (nb >= (1L << 31)) return -100;
  a.nblocks = (int)nb;
  *cfg = (rows << 16) | (bn << 4) | (rows == 256 ? 8 : 0) | (PREC_F16X3 << 1) | 4;
  if (rows == 256) {
    switch (bn) {
      case 128: launch_sp<128, 8>(a, s); break;
      case 192: launch_sp<192, 8>(a, s); break;
      case 256: launch_sp<256, 8>(a, s); break;
      default: return -100;
    }
    return 0;
  }
  switch (bn) {
    case 64: launch_sp<64, 4>(a, s); break;
    case 96: launch_sp<96, 4>(a, s); break;
    case 128: launch_sp<128, 4>(a, s); break;
    case 192: launch_sp<192, 4>(a, s); break;
    case 256: launch_sp<256, 4>(a, s); break;
    default: return -100;
  }
  return 0;
}

// 256 x BN tile, 8 waves (tile codes 12..14)
int launch_conv_ad8(ConvArgs& a, long Ml, int bn, int* cfg, hipStream_t s) {
  a.tiles_n = (int)((a.Cout + bn - 1) / bn);
  const long nb = ((Ml + 255) / 256) * a.tiles_n;
  if (nb >= (1L << 31)) return -100;
  a.nblocks = (int)nb;
  *cfg = (256 << 16) | (bn << 4) | 8 | (PREC_F16X3 << 1) | 4;
  switch (bn) {
    case 128: launch_bn<128, 8>(a, s); break;
    case 192: launch_bn<192, 8>(a, s); break;
    case 256: launch_bn<256, 8>(a, s); break;
    default: return -100;
  }
  return 0;
}

int launch_conv_ad(ConvArgs& a, long Ml, int force_bn_arg, int* cfg, hipStream_t s) {
  const long tm = (Ml + 127) / 128;
  const long t128 = (a.Cout + 127) / 128, t64 = (a.Cout + 63) / 64;
  static const int env_bn = getenv("MSPI_CONV_BN") ? atoi(getenv("MSPI_CONV_BN")) : 0;
  int bn = force_bn_arg ? force_bn_arg : env_bn;
  if (bn == 1) {   // one column tile holding every output channel: the activations are fetched exactly once
    bn = (a.Cout + 31) / 32 * 32;
    if (bn > 256) return -100;
  }
  if (bn == 0) bn = (t64 * 64 < t128 * 128 || tm * t128 < 384) ? 64 : 128;   // less padding, or a grid that fills the chip
  a.tiles_n = (int)((a.Cout + bn - 1) / bn);
  const long nb = tm * a.tiles_n;
  if (nb >= (1L << 31)) return -100;
  a.nblocks = (int)nb;
  *cfg = (128 << 16) | (bn << 4) | (PREC_F16X3 << 1) | 4;   // loader code 4 = LDS-DMA
  switch (bn) {
    case 32: launch_bn<32>(a, s); break;
    case 64: launch_bn<64>(a, s); break;
    case 96: launch_bn<96>(a, s); break;
    case 128: launch_bn<128>(a, s); break;
    case 160: launch_bn<160>(a, s); break;
    case 192: launch_bn<192>(a, s); break;
    case 224: launch_bn<224>(a, s); break;
    case 256: launch_bn<256>(a, s); break;
    default: return -100;
  }
  return 0;
}

}  // namespace mspi
