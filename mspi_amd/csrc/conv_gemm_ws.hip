// f16x3 GEMM on pre-split activation planes, 256 x 128 tile, one LOADER wave + eight MFMA waves (tile code 15).
//
// Round-3 reading of the 128 x 64 LDS-DMA kernel (profiles/r03_gemm_spr.txt): per 32-deep K step a
// workgroup moves 24 KB through the CU's one vector-memory path (64 B/clk) for 48 MFMAs -- 384 cycles of loads for 384 cycles
// of matrix work per CU, issued by the SAME waves in barrier-separated phases (a wave that issues a load while the path is busy
// stalls ~50 ns per instruction, in order, in front of its MFMAs), with one stage of cover against 1-1.5 us of loaded L2
// latency.  Matrix pipe 0.28, load path 0.3, both idle two thirds of the time.
//
// This kernel changes the three ratios at once:
//   * 256 x 128 tile, eight MFMA waves as 4 x 2 with a 64 x 64 register tile each (2 x 2 accumulators): 48 KB per step for
//     192 MFMAs -- 768 load cycles against 1536 matrix cycles per CU -- and 16 fragment reads per 24 MFMAs per wave;
//   * a NINTH wave does nothing but issue the LDS-DMA of both operands (48 x 1 KB per step): no MFMA wave ever issues a
//     vector-memory instruction, so none waits at the load path;
//   * three LDS stages (144 KB, one workgroup per CU): two steps of DMA in flight (~1.5 us of cover) behind the loader's
//     counted vmcnt; ONE barrier per step for all nine waves: arriving, the loader vouches for step k having landed and the
//     MFMA waves for step k - 1 being consumed, whose slot the loader refills with step k + 2 right behind the barrier.
// Staging layout / swizzle / K order / epilogue are conv_gemm_dma_kernel's: results are bit-identical to tile codes 6 .. 14.
#include "conv_common.h"

namespace mspi {

__device__ __attribute__((aligned(16))) float g_zero16_ws[4] = {0.f, 0.f, 0.f, 0.f};   // source of weight rows past Cout
typedef __attribute__((address_space(3))) void lds_void_w;

constexpr int WS_BM = 256, WS_BN = 128, WS_NST = 3;
constexpr int WS_A_BYTES = WS_BM * 64;             // one f16 activation plane of a step: 256 rows x 32 k
constexpr int WS_W_BYTES = WS_BN * 64;
constexpr int WS_STAGE = 2 * WS_A_BYTES + 2 * WS_W_BYTES;      // 48 KB
constexpr int WS_DMA = (2 * WS_A_BYTES + 2 * WS_W_BYTES) / 1024;   // 48 LDS-DMA instructions per step

__global__ __launch_bounds__(576, 1) void gemm_ws_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];     // WS_NST stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int logical = xcd_logical_block(blockIdx.x, p.nblocks);
  const int tile_n = logical % p.tiles_n, tile_m = logical / p.tiles_n;
  const int m0 = tile_m * WS_BM, n0 = tile_n * WS_BN;
  const int nk = (int)(p.ldw / BK);

  if (wave == 8) {
    // ------------------------------------------------------------------ loader
    // one LDS-DMA instruction = 16 rows x 64 B of a plane: row r = 16 g + lane/4, slot lane%4 holds segment slot ^ ((r>>2)&3)
    const int b_seg = (lane & 3) ^ ((lane >> 4) & 3);
    const int rsub = lane >> 2;
    const _Float16* wh = reinterpret_cast<const _Float16*>(p.w);
    const long wplane = (long)p.Cout * p.ldw;
    // per-lane source rows, fixed over K: 16 activation groups (rows past M repeat the last row: results dropped), 8 weight groups
    const _Float16* asrc[WS_BM / 16];
#pragma unroll
    for (int g = 0; g < WS_BM / 16; ++g) {
      int m = m0 + g * 16 + rsub;
      if (m >= p.M) m = p.M - 1;
      asrc[g] = p.xs + (long)m * p.ldxs + b_seg * 8;
    }
    const _Float16* wsrc[WS_BN / 16];
    bool wok[WS_BN / 16];
#pragma unroll
    for (int g = 0; g < WS_BN / 16; ++g) {
      const int n = n0 + g * 16 + rsub;
      wok[g] = n < p.Cout;
      wsrc[g] = wh + (long)(wok[g] ? n : 0) * p.ldw + b_seg * 8;
    }
    const long xplane = p.xplane;
    auto issue = [&](int slot, int k0) {
      unsigned char* base = ws_smem + slot * WS_STAGE;
#pragma unroll
      for (int g = 0; g < WS_BM / 16; ++g) {
        __builtin_amdgcn_global_load_lds(asrc[g] + k0, (lds_void_w*)(base + g * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(asrc[g] + xplane + k0, (lds_void_w*)(base + WS_A_BYTES + g * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < WS_BN / 16; ++g) {
        const void* s_hi = wok[g] ? (const void*)(wsrc[g] + k0) : (const void*)g_zero16_ws;
        const void* s_lo = wok[g] ? (const void*)(wsrc[g] + wplane + k0) : (const void*)g_zero16_ws;
        __builtin_amdgcn_global_load_lds(s_hi, (lds_void_w*)(base + 2 * WS_A_BYTES + g * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(s_lo, (lds_void_w*)(base + 2 * WS_A_BYTES + WS_W_BYTES + g * 1024), 16, 0, 0);
      }
    };
    issue(0, 0);
    if (nk > 1) issue(1, BK);
    for (int k = 0; k < nk; ++k) {
      // step k has landed: at most the one step issued after it (k + 1) may still be in flight
      if (k + 1 < nk) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                     // barrier k
      if (k + 2 < nk) issue((k + 2) % WS_NST, (k + 2) * BK);            // the slot of step k - 1, consumed before barrier k
    }
    return;
  }

  // -------------------------------------------------------------------- MFMA waves: 4 (rows) x 2 (columns), 64 x 64 each
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  v16f acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int k = 0; k < nk; ++k) {
    __builtin_amdgcn_s_barrier();                                       // barrier k: step k is in LDS
    const unsigned char* base = ws_smem + (k % WS_NST) * WS_STAGE;
    const _Float16* Ah = reinterpret_cast<const _Float16*>(base);
    const _Float16* Al = reinterpret_cast<const _Float16*>(base + WS_A_BYTES);
    const _Float16* Bh = reinterpret_cast<const _Float16*>(base + 2 * WS_A_BYTES);
    const _Float16* Bl = reinterpret_cast<const _Float16*>(base + 2 * WS_A_BYTES + WS_W_BYTES);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      v8h ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + li;
        const int o = r * 32 + ((((2 * lh + sub)) ^ ((r >> 2) & 3)) << 3);
        ah[i] = *reinterpret_cast<const v8h*>(&Ah[o]);
        al[i] = *reinterpret_cast<const v8h*>(&Al[o]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = wn * 64 + j * 32 + li;
        const int o = r * 32 + ((((2 * lh + sub)) ^ ((r >> 2) & 3)) << 3);
        bh[j] = *reinterpret_cast<const v8h*>(&Bh[o]);
        bl[j] = *reinterpret_cast<const v8h*>(&Bl[o]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (!kSingleProduct) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // my fragment reads of this slot are done before I arrive at the next barrier
  }

  // ---- epilogue (conv_gemm_dma_kernel's): C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  bool bad = false;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
      if (col >= p.Cout) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
      const int rb0 = m0 + wm * 64 + i * 32 + 4 * lh;
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
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const bool odd = li & 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          unsigned own[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int r = 2 * q + e;
            const float pre = acc[i][j][r] * p.out_scale + bv + rv[r];
            bad |= rb0 + (r & 3) + 8 * (r >> 2) < p.M && nonfinite(pre);
            const float v = act_apply(pre, p.act);
            _Float16 h, l;
            split_f16(v, h, l);
            h2 pr = {h, l};
            own[e] = __builtin_bit_cast(unsigned, pr);
          }
          const unsigned got = (unsigned)__shfl_xor((int)(odd ? own[0] : own[1]), 1, 64);
          const unsigned mine = odd ? own[1] : own[0];
          const h2 a = __builtin_bit_cast(h2, odd ? got : mine), b = __builtin_bit_cast(h2, odd ? mine : got);   // columns c0, c0+1
          const int r = 2 * q + (odd ? 1 : 0);
          const int row = rb0 + (r & 3) + 8 * (r >> 2);
          const int c0 = col & ~1;
          if (row < p.M) {
            h2 hi = {a[0], b[0]}, lo = {a[1], b[1]};
            *reinterpret_cast<h2*>(p.ys + (long)row * p.ldys + c0) = hi;
            *reinterpret_cast<h2*>(p.ys + p.yplane + (long)row * p.ldys + c0) = lo;
          }
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb0 + (r & 3) + 8 * (r >> 2);
        const float pre = acc[i][j][r] * p.out_scale + bv + rv[r];
        bad |= row < p.M && nonfinite(pre);
        if (row < p.M) p.y[(long)row * p.ldy + col] = act_apply(pre, p.act);
      }
    }
  report_nonfinite(p.status, bad);
}

// tile code 15: 256 x 128, loader wave + 8 MFMA waves, 144 KB of LDS
int launch_conv_ws(ConvArgs& a, long Ml, int* cfg, hipStream_t s) {
  a.tiles_n = (int)((a.Cout + WS_BN - 1) / WS_BN);
  const long nb = ((Ml + WS_BM - 1) / WS_BM) * a.tiles_n;
  if (nb >= (1L << 31)) return -100;
  a.nblocks = (int)nb;
  *cfg = (WS_BM << 16) | (WS_BN << 4) | 8 | (PREC_F16X3 << 1) | 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WS_NST * WS_STAGE) != hipSuccess) {
      (void)hipGetLastError();
      return -100;
    }
    attr = true;
  }
  hipLaunchKernelGGL(gemm_ws_kernel, dim3(a.nblocks), dim3(576), WS_NST * WS_STAGE, s, a);
  return 0;
}

}  // namespace mspi
