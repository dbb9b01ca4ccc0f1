// f16x3 GEMM on pre-split activation planes, A operand global -> registers, deep weight ring (tile codes 15 / 17 / 18).
//
// Where conv_gemm_dma_kernel<.., APRE> loses its time (round 3 reading of the round-2 numbers): its two LDS stages hold A and W
// (24 KB per 128 x 64 x 32 step), so exactly ONE stage of DMA is in flight while a stage is computed -- 12 MFMAs per wave, 0.2-0.4
// us -- against a loaded L2 -> LDS latency of 1-1.5 us.  25088 x 1536 x 384 at 129 us is 48 k-steps of ~1.2 us per workgroup:
// every step waits for its data; the matrix pipe (0.28 busy) and the L1 path (10.5 of ~34 TB/s) both idle in between.  A third
// stage cost a resident workgroup (LDS) and one more stage is not enough anyway.
//
// Here the A operand never touches LDS: a wave owns 32 rows of the tile, and its MFMA fragments (lane (i, h): row i,
// k = 16 h + 8 sub .. + 7 of a 32-deep step) are 16-B runs of the hi / lo planes, loaded straight into registers -- four
// loads per step and wave, kept in a ring of NST register stages.  Only the weight tile goes through LDS (8 KB per step at
// BN = 64), so the LDS-DMA ring is NST = 6 / 4 / 3 stages deep in 48-72 KB and NST - 1 steps of loads are in flight behind a
// COUNTED vmcnt (loads and LDS-DMA retire in issue order) + raw s_barrier.  Same staging layout, swizzle, K order and epilogue as
// conv_gemm_dma_kernel: results are bit-identical to tile codes 6 .. 14.
#include "conv_common.h"
#include <stdlib.h>

namespace mspi {

__device__ __attribute__((aligned(16))) float g_zero16_spr[4] = {0.f, 0.f, 0.f, 0.f};   // source of weight rows past Cout
typedef __attribute__((address_space(3))) void lds_void_r;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BN, int NST>
__global__ __launch_bounds__(256, 2) void gemm_spr_kernel(const ConvArgs p) {
  constexpr int TN = BN / 32;
  constexpr int P_BYTES = BN * 32 * 2;          // one f16 weight plane of a step: BN rows x 32 k
  constexpr int STAGE = 2 * P_BYTES;
  constexpr int HBI = BN / 64;                  // weight DMA instruction pairs per wave and step (16 rows each)
  constexpr int VPS = 2 * HBI + 4;              // vector-memory operations per wave and step: weight DMA + 4 A-fragment loads
  static_assert(BN % 64 == 0 && BN <= 256 && NST >= 3 && VPS * (NST - 2) <= 63, "tile");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int logical = xcd_logical_block(blockIdx.x, p.nblocks);
  const int tile_n = logical % p.tiles_n, tile_m = logical / p.tiles_n;
  const int m0 = tile_m * 128, n0 = tile_n * BN;

  // A fragments: row m0 + wave*32 + li (rows past M repeat the last row; their results are dropped), 16 halves from k0 + 16 lh
  int arow = m0 + wave * 32 + li;
  if (arow >= p.M) arow = p.M - 1;
  // p.dbg & 128 (MSPI_SPR_FAKE=1, timing experiment only, results are garbage): the ACCESS PATTERN of a fragment-major plane
  // layout -- every load instruction reads 1 KB of contiguous memory -- on the row-major data
  const bool fake = p.dbg & 128;
  const long fk8 = p.ldxs / 8;
  const _Float16* ah_p = fake ? p.xs + (((long)(arow >> 5) * fk8 + lh) * 32 + li) * 8 : p.xs + (long)arow * p.ldxs + 16 * lh;
  const _Float16* al_p = ah_p + p.xplane;
  // weight DMA: plane chunks (i*4+wave)*64 + lane: row r = q/4, slot q%4, segment = slot ^ ((r>>2)&3)   (as conv_gemm_ad.hip)
  const _Float16* wh = reinterpret_cast<const _Float16*>(p.w);
  const long wplane = (long)p.Cout * p.ldw;
  const int b_seg = (lane & 3) ^ ((lane >> 4) & 3);
  const _Float16* wsrc[HBI];
  bool wok[HBI];
#pragma unroll
  for (int i = 0; i < HBI; ++i) {
    const int n = n0 + (i * 4 + wave) * 16 + (lane >> 2);
    wok[i] = n < p.Cout;
    wsrc[i] = fake ? wh + ((long)((n0 >> 4) + i * 4 + wave) * (p.ldw / 32)) * 512 + lane * 8 : wh + (long)(wok[i] ? n : 0) * p.ldw + b_seg * 8;
  }

  // The A loads are inline asm: the compiler's own wait analysis merges the prologue's and the back edge's scoreboards at the loop
  // header into vmcnt(0) before the first fragment use of every trip (a full drain of the ring every NST steps).  Invisible to it,
  // the loads are covered by the counted waits below alone -- every step issues exactly VPS operations (steps past the end of K
  // read the 16-byte zero page: one hot line, no bandwidth), so one immediate fits every step.
  v8h ah[NST][2], al[NST][2];
#define SPR_ISSUE(slot_, k0_, live_)                                                                                     \
  do {                                                                                                                   \
    unsigned char* base_ = smem + (slot_) * STAGE;                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < HBI; ++i_) {                                                                 \
      const bool ok_ = (live_) && wok[i_];                                                                               \
      const long kw_ = fake ? (long)(k0_) * 16 : (long)(k0_);                                                            \
      const void* s_hi_ = ok_ ? (const void*)(wsrc[i_] + kw_) : (const void*)g_zero16_spr;                               \
      const void* s_lo_ = ok_ ? (const void*)(wsrc[i_] + wplane + kw_) : (const void*)g_zero16_spr;                      \
      __builtin_amdgcn_global_load_lds(s_hi_, (lds_void_r*)(base_ + (i_ * 4 + wave) * 1024), 16, 0, 0);                  \
      __builtin_amdgcn_global_load_lds(s_lo_, (lds_void_r*)(base_ + P_BYTES + (i_ * 4 + wave) * 1024), 16, 0, 0);        \
    }                                                                                                                    \
    const long sub1_ = fake ? 512 : 8;                      /* halves from the sub 0 to the sub 1 fragment */             \
    const long ka_ = fake ? (long)(k0_) * 32 : (long)(k0_);                                                              \
    const _Float16* pa_ = (live_) ? ah_p + ka_ : reinterpret_cast<const _Float16*>(g_zero16_spr);                        \
    const _Float16* pb_ = (live_) ? al_p + ka_ : reinterpret_cast<const _Float16*>(g_zero16_spr);                        \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ah[slot_][0]) : "v"(pa_) : "memory");                         \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ah[slot_][1]) : "v"((live_) ? pa_ + sub1_ : pa_) : "memory"); \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(al[slot_][0]) : "v"(pb_) : "memory");                         \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(al[slot_][1]) : "v"((live_) ? pb_ + sub1_ : pb_) : "memory"); \
  } while (0)

  v16f acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int nk = (int)(p.ldw / BK);             // >= NST - 1 (checked on the host)
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) SPR_ISSUE(s, s * BK, true);
  __builtin_amdgcn_sched_barrier(0);

#define SPR_COMPUTE(u_)                                                                                                 \
  do {                                                                                                                   \
    const _Float16* Bh_ = reinterpret_cast<const _Float16*>(smem + (u_) * STAGE);                                        \
    const _Float16* Bl_ = reinterpret_cast<const _Float16*>(smem + (u_) * STAGE + P_BYTES);                              \
    _Pragma("unroll") for (int sub_ = 0; sub_ < 2; ++sub_) {                                                             \
      v8h bh_[TN], bl_[TN];                                                                                              \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) {                                                                \
        const int r_ = j_ * 32 + li;                                                                                     \
        const int o_ = r_ * 32 + ((((2 * lh + sub_)) ^ ((r_ >> 2) & 3)) << 3);                                           \
        bh_[j_] = *reinterpret_cast<const v8h*>(&Bh_[o_]);                                                               \
        bl_[j_] = *reinterpret_cast<const v8h*>(&Bl_[o_]);                                                               \
      }                                                                                                                  \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) {                                                                \
        if (!kSingleProduct) {                                                                                           \
          acc[j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u_][sub_], bh_[j_], acc[j_], 0, 0, 0);                     \
          acc[j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u_][sub_], bl_[j_], acc[j_], 0, 0, 0);                     \
        }                                                                                                                \
        acc[j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u_][sub_], bh_[j_], acc[j_], 0, 0, 0);                       \
      }                                                                                                                  \
    }                                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* my fragment reads of this slot are done before the next barrier */ \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  } while (0)

  // NST steps per trip, NO branch inside a trip -- every step waits for its own data with NST - 2 newer steps
  // still in flight, passes the barrier (everybody's step `it` has landed, everybody is done with step it - 1), refills the
  // slot of step it - 1 with step it + NST - 1 and computes.  (With `if (it < nk)` around the steps the accumulators went
  // through PHI copies behind every MFMA group.)
  // K is processed in whole trips: the steps past its end were issued from the zero page (zero weights, zero activations) and add
  // nothing -- a peeled, conditional tail would put the asynchronously loaded fragment registers through PHI copies at its joins,
  // i.e. copy them before their data has landed.
  // diagnostic stamps (mspi_debug_stamps, tools/spr_stamps.py): p.ws != NULL only then.  One lane per workgroup, the second trip.
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.ws);
#define SPR_STAMP(i_)                                                                                                   \
  do {                                                                                                                  \
    if (stamps && tid == 0 && it0 == NST) stamps[(long)blockIdx.x * 64 + u * 4 + (i_)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#pragma unroll 1
  for (int it0 = 0; it0 < nk; it0 += NST) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      SPR_STAMP(0);
      wait_vmcnt<VPS * (NST - 2)>();
      SPR_STAMP(1);
      __builtin_amdgcn_s_barrier();
      SPR_STAMP(2);
      SPR_ISSUE((u + NST - 1) % NST, (it0 + u + NST - 1) * BK, it0 + u + NST - 1 < nk);
      __builtin_amdgcn_sched_barrier(0);         // the loads are issued HERE (the scheduler would sink them below the MFMAs)
      SPR_STAMP(3);
      SPR_COMPUTE(u);
    }
  }
#undef SPR_STAMP
  wait_vmcnt<0>();                               // the zero-page loads of the last steps
#undef SPR_COMPUTE
#undef SPR_ISSUE

  // ---- epilogue (conv_gemm_dma_kernel's): C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  bool bad = false;
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
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const bool odd = li & 1;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        unsigned own[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int r = 2 * q + e;
          const float pre = acc[j][r] * p.out_scale + bv + rv[r];
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
      const float pre = acc[j][r] * p.out_scale + bv + rv[r];
      bad |= row < p.M && nonfinite(pre);
      if (row < p.M) p.y[(long)row * p.ldy + col] = act_apply(pre, p.act);
    }
  }
  report_nonfinite(p.status, bad);
}

// tile codes 15 / 17 / 18: 128 x {64, 128, 192}; returns -100 when the shape is outside the kernel (few k-steps)
unsigned long long* g_spr_stamps = nullptr;   // mspi_debug_stamps

int launch_conv_spr(ConvArgs& a, long Ml, int bn, int* cfg, hipStream_t s) {
  a.ws = reinterpret_cast<float*>(g_spr_stamps);
  static const int fake = getenv("MSPI_SPR_FAKE") ? atoi(getenv("MSPI_SPR_FAKE")) : 0;
  if (fake) a.dbg |= 128;
  const int nst = bn == 64 ? 6 : (bn == 128 ? 4 : 3);
  if (a.ldw / BK < nst - 1) return -100;
  a.tiles_n = (int)((a.Cout + bn - 1) / bn);
  const long nb = ((Ml + 127) / 128) * a.tiles_n;
  if (nb >= (1L << 31)) return -100;
  a.nblocks = (int)nb;
  *cfg = (128 << 16) | (bn << 4) | (PREC_F16X3 << 1) | 4;
  const dim3 g(a.nblocks), b(256);
  switch (bn) {
    case 64: hipLaunchKernelGGL((gemm_spr_kernel<64, 6>), g, b, 0, s, a); break;
    case 128: hipLaunchKernelGGL((gemm_spr_kernel<128, 4>), g, b, 0, s, a); break;
    case 192: hipLaunchKernelGGL((gemm_spr_kernel<192, 3>), g, b, 0, s, a); break;
    default: return -100;
  }
  return 0;
}

}  // namespace mspi
