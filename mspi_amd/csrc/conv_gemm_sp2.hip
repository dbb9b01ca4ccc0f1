// f16x3 GEMM on pre-split planes, 2 x 2 waves with a (32 TMW) x (32 TNW) register tile per wave.
//
// conv_gemm_dma_kernel (conv_gemm_ad.hip) gives each of its 4 or 8 waves a 32 x BN slab: every wave re-reads the whole
// weight tile from LDS, and per staged byte the 128 x 64 workgroup tile does 65 (issued) flops -- at the 0.4 MFMA-pipe target
// that is 15 TB/s of L2 -> LDS fill, which the chip does not have beside the LDS reads.  Here both operands are pre-split
// planes (mspi_gemm_sp_fwd: the ConvNeXt / Swin / MViT fc1, fc2, qkv layers), the workgroup tile is (64 TMW) x (64 TNW)
// (128 x 128: 98 flops per staged byte; 128 x 256: 131) and a wave reads only its own TMW + TNW fragments per k16 step for
// TMW * TNW * 3 MFMAs (2 x 2: 8 ds_read_b128 per 12 MFMAs instead of 6 per 6).  NST-deep LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4, source-side XOR swizzle, zero page for rows past M / Cout); NST = 3 keeps two K steps of DMA
// in flight behind a counted vmcnt and a raw s_barrier.
#include "conv_common.h"

namespace mspi {

__device__ __attribute__((aligned(16))) float g_zero16_sp2[4] = {0.f, 0.f, 0.f, 0.f};
typedef __attribute__((address_space(3))) void lds_void2;

template <int TMW, int TNW, int NST>
__global__ __launch_bounds__(256, (64 * (TMW + TNW) * 128 * NST <= 80 * 1024) ? 2 : 1) void gemm_sp2_kernel(const ConvArgs p) {
  constexpr int BM = 64 * TMW, BN = 64 * TNW;
  constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;      // bytes: rows x 32 halves
  constexpr int STAGE = 2 * (A_PLANE + B_PLANE);
  constexpr int GA = BM / 16, GB = BN / 16;                // 1-KB DMA groups (16 rows) per plane
  constexpr int DMA_PER_STAGE = 2 * (GA + GB) / 4;         // per thread, identical for all threads
  static_assert((2 * (GA + GB)) % 4 == 0, "DMA instructions must divide evenly over the four waves");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int logical = xcd_logical_block(blockIdx.x, p.nblocks);
  const int tile_n = logical % p.tiles_n, tile_m = logical / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const _Float16* wh = reinterpret_cast<const _Float16*>(p.w);
  const long wplane = (long)p.Cout * p.ldw;
  const int seg = (lane & 3) ^ ((lane >> 4) & 3);          // source segment of the slot this lane fills (row = 16 g + lane / 4)

  auto issue_stage = [&](int st, int k0) {
    unsigned char* base = smem + st * STAGE;
    // instruction index ii = 4 * i + wave runs over [A hi | A lo | B hi | B lo] groups
#pragma unroll
    for (int i = 0; i < DMA_PER_STAGE; ++i) {
      const int ii = 4 * i + wave;
      const void* src;
      unsigned char* dst;
      if (ii < 2 * GA) {
        const int pl = ii / GA, g = ii - pl * GA;
        const int m = m0 + g * 16 + (lane >> 2);
        const bool ok = m < p.M;
        src = ok ? (const void*)(p.xs + (pl ? p.xplane : 0) + (long)m * p.ldxs + k0 + seg * 8) : (const void*)g_zero16_sp2;
        dst = base + pl * A_PLANE + g * 1024;
      } else {
        const int jj = ii - 2 * GA;
        const int pl = jj / GB, g = jj - pl * GB;
        const int n = n0 + g * 16 + (lane >> 2);
        const bool ok = n < p.Cout;
        src = ok ? (const void*)(wh + (pl ? wplane : 0) + (long)n * p.ldw + k0 + seg * 8) : (const void*)g_zero16_sp2;
        dst = base + 2 * A_PLANE + pl * B_PLANE + g * 1024;
      }
      __builtin_amdgcn_global_load_lds(src, (lds_void2*)dst, 16, 0, 0);
    }
  };

  v16f acc[TMW][TNW];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  struct Frag { v8h ah[TMW], al[TMW], bh[TNW], bl[TNW]; };
  auto frag_read = [&](Frag& f, int st, int sub) {
    const unsigned char* base = smem + st * STAGE;
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
      const int r = wm * 32 * TMW + i * 32 + li;
      const int o = (r * 32 + (((2 * lh + sub) ^ ((r >> 2) & 3)) << 3)) * 2;
      f.ah[i] = *reinterpret_cast<const v8h*>(base + o);
      f.al[i] = *reinterpret_cast<const v8h*>(base + A_PLANE + o);
    }
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int r = wn * 32 * TNW + j * 32 + li;
      const int o = (r * 32 + (((2 * lh + sub) ^ ((r >> 2) & 3)) << 3)) * 2;
      f.bh[j] = *reinterpret_cast<const v8h*>(base + 2 * A_PLANE + o);
      f.bl[j] = *reinterpret_cast<const v8h*>(base + 2 * A_PLANE + B_PLANE + o);
    }
  };
  auto frag_mfma = [&](const Frag& f) {
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
        if (!kSingleProduct) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
      }
  };

  const int nk = (int)(p.ldw / BK);
  Frag f0, f1;
  issue_stage(0, 0);
  if (NST == 3 && nk > 1) issue_stage(1, BK);
  if (NST == 3 && nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_STAGE) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int cur = 0;
  for (int it = 0; it < nk; ++it) {
    if (NST == 2) {
      if (it + 1 < nk) issue_stage(cur ^ 1, (it + 1) * BK);
    } else {
      const int nxt2 = cur >= 1 ? cur - 1 : 2;   // (cur + 2) % 3
      if (it + 2 < nk) issue_stage(nxt2, (it + 2) * BK);
    }
    frag_read(f0, cur, 0);
    frag_read(f1, cur, 1);
    frag_mfma(f0);
#pragma unroll
    for (int g = 0; g < 2 * (TMW + TNW); ++g) {              // sub-step 1's fragment reads spread under sub-step 0's MFMAs
      __builtin_amdgcn_sched_group_barrier(0x008, (3 * TMW * TNW) / (2 * (TMW + TNW)) > 0 ? (3 * TMW * TNW) / (2 * (TMW + TNW)) : 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    frag_mfma(f1);
    if (NST == 3 && it + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_STAGE) : "memory");   // stage it+1 landed, it+2 may fly
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (NST == 2) cur ^= 1;
    else cur = cur == 2 ? 0 : cur + 1;
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  bool bad = false;
#pragma unroll
  for (int j = 0; j < TNW; ++j) {
    const int col = n0 + wn * 32 * TNW + j * 32 + li;
    if (col >= p.Cout) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
      const int rb0 = m0 + wm * 32 * TMW + i * 32 + 4 * lh;
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
      if (p.ys) {   // split-plane output (see conv_gemm_ad.hip): lanes c and c+1 trade halves, every store is a 4-B pair
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const bool odd = li & 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          unsigned own[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int r = 2 * q + e;
            const float pre = acc[i][j][r] * p.out_scale + bv + rv[r];
            bad |= nonfinite(pre);
            _Float16 h, l;
            split_f16(act_apply(pre, p.act), h, l);
            h2 pr = {h, l};
            own[e] = __builtin_bit_cast(unsigned, pr);
          }
          const unsigned got = (unsigned)__shfl_xor((int)(odd ? own[0] : own[1]), 1, 64);
          const unsigned mine = odd ? own[1] : own[0];
          const h2 a = __builtin_bit_cast(h2, odd ? got : mine), b = __builtin_bit_cast(h2, odd ? mine : got);
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
        bad |= nonfinite(pre);
        if (row < p.M) p.y[(long)row * p.ldy + col] = act_apply(pre, p.act);
      }
    }
  }
  report_nonfinite(p.status, bad);
}

// variant: 0 128x128 ring 2, 1 128x128 ring 3, 2 128x256 ring 2, 3 256x128 ring 2
int launch_conv_sp2(ConvArgs& a, long Ml, int variant, int* cfg, hipStream_t s) {
  const int bm = variant == 3 ? 256 : 128, bn = variant == 2 ? 256 : 128;
  a.tiles_n = (int)((a.Cout + bn - 1) / bn);
  const long nb = ((Ml + bm - 1) / bm) * a.tiles_n;
  if (nb >= (1L << 31)) return -100;
  a.nblocks = (int)nb;
  *cfg = (bm << 16) | (bn << 4) | (PREC_F16X3 << 1) | 4 | 8;   // 8: the 2 x 2-wave form (mspi_conv_last_config)
  const dim3 g(a.nblocks), b(256);
  switch (variant) {
    case 0: hipLaunchKernelGGL((gemm_sp2_kernel<2, 2, 2>), g, b, 0, s, a); break;
    case 1: hipLaunchKernelGGL((gemm_sp2_kernel<2, 2, 3>), g, b, 0, s, a); break;
    case 2: hipLaunchKernelGGL((gemm_sp2_kernel<2, 4, 2>), g, b, 0, s, a); break;
    case 3: hipLaunchKernelGGL((gemm_sp2_kernel<4, 2, 2>), g, b, 0, s, a); break;
    default: return -100;
  }
  return 0;
}

}  // namespace mspi
