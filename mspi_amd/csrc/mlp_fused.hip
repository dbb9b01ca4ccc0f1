// Fused channel MLP:  y = res + fc2( act( fc1( LayerNorm(x) ) ) )   on rows of a channels-last matrix.
//
// The 4C-wide hidden activation of a ConvNeXt / transformer MLP never leaves the CU: for C = 96 the unfused pair
// (fc1+GELU, fc2+residual) moves 4.2x the bytes of this kernel and both halves are HBM-bound (profiles/r01).
//
// Shape of the computation (f16x3 split products on v_mfma_f32_32x32x16_f16, fp32 accumulate -- see conv_gemm.hip):
//   * a wave owns 32*TM complete rows.  It loads them once, applies the LayerNorm in registers (the two lanes that
//     share a row exchange partial sums with one DPP/permute), splits them into f16 hi/lo and KEEPS them as MFMA
//     B-operand fragments for the whole kernel (C/16 * 8 VGPRs per 32 rows).
//   * the hidden dimension is walked in chunks of 32.  Phase 1 computes the TRANSPOSED chunk
//         H^T[32 hidden][32 rows] = W1[chunk rows][:] . X^T            (A operand = weights, B operand = X)
//     so that the accumulator registers of a lane (one row m = lane%32, 16 hidden units) are, after bias + GELU + split,
//     directly the B operand of phase 2
//         Y^T[C][32 rows] += W2[:, chunk] . H                          (A operand = weights, B operand = H)
//     The k-slot -> hidden-unit permutation this implies is absorbed into the (offline) packing of W2.
//   * the chunk loop is a software pipeline: iteration j holds phase 2 of chunk j-1, the GELU + split of chunk j and phase 1
//     of chunk j+1 in one hand-interleaved instruction stream (a weight triple of 3 MFMAs in front of each of the 16
//     accumulator elements' ~20 VALU instructions).  A wave's own VALU work issues in the shadow of its MFMAs at no cost;
//     an MFMA-phase wave beside a VALU-phase wave on one SIMD slows both (tools/coissue_probe.hip).  Round 3, batch-8
//     shapes: C = 192 340 -> 302 us, C = 96 303 -> 294 us (that one is bound by the GELU's VALU issue: 390 instructions per
//     chunk and wave against 36 MFMAs).
//   * weights are packed in exact fragment order (engine.pack_mlp), so a chunk's weights are one linear 24/48 KB
//     block: staged by LDS-DMA (global_load_lds, 1 KB per wave instruction, no swizzle needed) into a ring of NS
//     stages, read back with conflict-free ds_read_b128 (lane*16 B).  No activation ever goes through LDS.
//   * epilogue: Y^T accumulators -> +bias, +residual -> 16-B stores (4 consecutive channels per lane).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace mspi {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

struct MlpArgs {
  const float* x; const float* gamma; const float* beta; const unsigned char* wp; const float* b1; const float* b2;
  const float* res; float* y;
  long M, ldx, ldr, ldy;
  int nch;            // hidden / 32
  int ln, act;
  float eps, inv_s1, inv_s2;
  int* status;
};

// Packed weight stage (one hidden chunk j of 32 units), all f16, 1 KB = 64 lanes x 8 halves per fragment:
//   W1 part: [ks = 0..C/16)[plane hi,lo][lane][e]   = W1s[j*32 + lane%32][16*ks + 8*(lane/32) + e]
//   W2 part: [s = 0..2)[ct = 0..C/32)[plane][lane][e] = W2s[ct*32 + lane%32][j*32 + (2*s + e/4)*8 + 4*(lane/32) + e%4]
// NWV waves per workgroup (4 or 8) share one weight ring: 8 waves (256 rows) halve the weight stream per row -- every workgroup
// pulls the WHOLE layer pair through LDS-DMA (1.15 MB at C = 192), and that issue is what the 4-wave loop spends most of its
// non-MFMA time on.
template <int C, int TM, int NS, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, (C == 96 && TM == 1 && NWV == 4) ? 2 : 1) void mlp_fused_kernel(const MlpArgs p) {
  constexpr int KS = C / 16, CT = C / 32;
  constexpr int W1B = KS * 2048, W2B = 2 * CT * 2048, SB = W1B + W2B;   // bytes per stage
  constexpr int DPW = SB / (1024 * NWV);                                // DMA instructions per wave per stage
  constexpr int BM = NWV * 32 * TM;
  static_assert(SB % (1024 * NWV) == 0, "a stage is dealt in whole 1-KB pieces to the waves");
  static_assert(C % 32 == 0 && SB % 4096 == 0, "C: multiple of 32");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * SB + (C == 96 ? 2048 : 4096)];   // the ring, then b1 (hidden <= 1024 floats)
  float* b1s = reinterpret_cast<float*>(smem + NS * SB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long row0 = (long)blockIdx.x * BM + wave * (32 * TM);

  // Software pipeline over the hidden chunks.  Iteration j holds, in ONE scheduling region of the wave,
  //     phase 2 of chunk j-1 (MFMA)  |  GELU + split of chunk j (VALU, ~24 instructions per hidden unit and lane)  |  phase 1 of chunk j+1 (MFMA)
  // because on this chip a wave's own VALU instructions issue for free in the shadow of its MFMAs, while an MFMA-phase wave
  // and a VALU-phase wave sharing a SIMD slow each other to less than the sum (tools/coissue_probe.hip: MFMA + 8 FMA
  // interleaved in one wave 42 cycles per pair = the VALU's own 41.5; split over two waves of a SIMD 78 / 110).
  // Ring stage j therefore holds W2 of chunk j and W1 of chunk j + 2 and is consumed by iteration j + 1.  The packing is
  // unchanged -- the W1 half of a stage is fetched from two chunks on (the last stages re-fetch chunk 0's, unused, so that
  // every wave's DMA count stays the same).
  static_assert(W1B % (DPW * 1024) == 0, "the W1 / W2 boundary of a stage falls between waves");
  auto issue_stage = [&](int j) {
    const bool w1 = wave * DPW * 1024 < W1B;
    const int jc = w1 ? (j + 2 < p.nch ? j + 2 : 0) : j;
    const unsigned char* src = p.wp + (long)jc * SB + (wave * DPW) * 1024 + lane * 16;
    unsigned char* dst = smem + (j % NS) * SB + (wave * DPW) * 1024;
#pragma unroll
    for (int d = 0; d < DPW; ++d)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + d * 1024), (lds_void*)(dst + d * 1024), 16, 0, 0);
  };
  // W1 of chunks 0 and 1 go to the two halves of the last slot (W1B == W2B), which the ring first overwrites at the top of
  // iteration 1, after both have been consumed (prologue, iteration 0)
  static_assert(W1B == W2B, "the prologue parks W1 of chunk 1 in a W2 half");
  constexpr int DP0 = W1B / (1024 * NWV);
  static_assert(W1B % (1024 * NWV) == 0, "W1 of a chunk is dealt in whole pieces");
  auto issue_w1_01 = [&]() {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int jc = c < p.nch ? c : 0;
#pragma unroll
      for (int d = 0; d < DP0; ++d)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(p.wp + (long)jc * SB + (wave * DP0 + d) * 1024 + lane * 16),
                                         (lds_void*)(smem + (NS - 1) * SB + c * W1B + (wave * DP0 + d) * 1024), 16, 0, 0);
    }
  };

  // ---- rows: load (clamped), LayerNorm, split.  lane (li, lh) holds row li of each row group, k = 16ks + 8lh + e
  float xr[TM][KS][8];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    long row = row0 + t * 32 + li;
    if (row >= p.M) row = p.M - 1;
    const float* xp = p.x + row * p.ldx + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 a = *reinterpret_cast<const float4*>(xp + 16 * ks);
      const float4 b = *reinterpret_cast<const float4*>(xp + 16 * ks + 4);
      xr[t][ks][0] = a.x; xr[t][ks][1] = a.y; xr[t][ks][2] = a.z; xr[t][ks][3] = a.w;
      xr[t][ks][4] = b.x; xr[t][ks][5] = b.y; xr[t][ks][6] = b.z; xr[t][ks][7] = b.w;
    }
  }
  // weight ring prologue + bias staging fly under the LayerNorm
  static_assert(NS >= 3, "the pipeline keeps stage j-1 in use while stage j is in flight and the prologue's slot is live");
  issue_w1_01();
  for (int j = 0; j < NS - 1; ++j) issue_stage(j < p.nch ? j : 0);     // always NS-1 stages: uniform DMA counts
  for (int i = tid; i < p.nch * 32; i += 64 * NWV) b1s[i] = p.b1[i];

  if (p.ln) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += xr[t][ks][e];
      s += __shfl_xor(s, 32, 64);
      const float mean = s / (float)C;
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = xr[t][ks][e] - mean; q = fmaf(d, d, q); }
      q += __shfl_xor(q, 32, 64);
      const float rstd = rsqrtf(q / (float)C + p.eps);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) xr[t][ks][e] = (xr[t][ks][e] - mean) * rstd;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + 16 * ks + 8 * lh);
      const float4 g1 = *reinterpret_cast<const float4*>(p.gamma + 16 * ks + 8 * lh + 4);
      const float4 c0 = *reinterpret_cast<const float4*>(p.beta + 16 * ks + 8 * lh);
      const float4 c1 = *reinterpret_cast<const float4*>(p.beta + 16 * ks + 8 * lh + 4);
      const float g8[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float c8[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) xr[t][ks][e] = fmaf(xr[t][ks][e], g8[e], c8[e]);
    }
  }
  v8h xh[TM][KS], xl[TM][KS];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        _Float16 h, l;
        split_f16(xr[t][ks][e], h, l);
        xh[t][ks][e] = h; xl[t][ks][e] = l;
      }

  v16f o[TM][CT];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[t][ct][i] = 0.f;

  auto phase1 = [&](const unsigned char* w1, v16f (&h)[TM]) {
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) h[t][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const v8h wh = *reinterpret_cast<const v8h*>(w1 + (ks * 2 + 0) * 1024);
      const v8h wl = *reinterpret_cast<const v8h*>(w1 + (ks * 2 + 1) * 1024);
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (!kSingleProduct) {
          h[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[t][ks], h[t], 0, 0, 0);
          h[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[t][ks], h[t], 0, 0, 0);
        }
        h[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[t][ks], h[t], 0, 0, 0);
      }
    }
  };
  // One pipeline iteration, hand-interleaved: the GELU + split of chunk j is cut into its 16 accumulator elements, and in
  // front of element g sit the weight triples (3 MFMAs on one pair of hi/lo weight fragments) [g*NTR/16, (g+1)*NTR/16) of
  //   phase 2 of chunk j-1 (triples 0 .. 2CT-1: (s, ct))   and   phase 1 of chunk j+1 (triples 2CT .. 2CT+KS-1: ks),
  // each group closed by a sched_barrier so the order survives the compiler.  The fragments of the next triple are read
  // from LDS one group ahead.
  v16f h[TM], hn[TM];
  v8h hh[TM][2], hl[TM][2], gh[TM][2], gl[TM][2];
  auto body = [&](auto has_p2, auto has_p1, int j, const unsigned char* st) {
    constexpr bool P2 = decltype(has_p2)::value, P1 = decltype(has_p1)::value;
    constexpr int T0 = P2 ? 0 : 2 * CT, T1 = P1 ? 2 * CT + KS : 2 * CT, NTR = T1 - T0;
    auto frag = [&](int tr, int plane) {
      const unsigned char* base = tr < 2 * CT ? st + W1B + (tr * 2) * 1024 : st + ((tr - 2 * CT) * 2) * 1024;
      return *reinterpret_cast<const v8h*>(base + plane * 1024);
    };
    float bv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = *reinterpret_cast<const float4*>(b1s + j * 32 + q * 8 + 4 * lh);
      bv[q * 4 + 0] = b.x; bv[q * 4 + 1] = b.y; bv[q * 4 + 2] = b.z; bv[q * 4 + 3] = b.w;
    }
    v8h wh, wl, nh, nl;
    if (NTR > 0) { wh = frag(T0, 0); wl = frag(T0, 1); }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
#pragma unroll
      for (int tr = T0 + g * NTR / 16; tr < T0 + (g + 1) * NTR / 16; ++tr) {
        if (tr + 1 < T1) { nh = frag(tr + 1, 0); nl = frag(tr + 1, 1); }
        __builtin_amdgcn_sched_barrier(0);      // the reads stay HERE: a whole group ahead of the MFMAs that wait for them
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          if (tr < 2 * CT) {
            const int s = tr / CT, ct = tr % CT;
            if (!kSingleProduct) {
              o[t][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hh[t][s], o[t][ct], 0, 0, 0);
              o[t][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hl[t][s], o[t][ct], 0, 0, 0);
            }
            o[t][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hh[t][s], o[t][ct], 0, 0, 0);
          } else {
            const int ks = tr - 2 * CT;
            const v16f zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (!kSingleProduct) {
              hn[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[t][ks], ks == 0 ? zero : hn[t], 0, 0, 0);   // inline-constant C: no zero fill
              hn[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[t][ks], hn[t], 0, 0, 0);
              hn[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[t][ks], hn[t], 0, 0, 0);
            } else {
              hn[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[t][ks], ks == 0 ? zero : hn[t], 0, 0, 0);
            }
          }
        }
        wh = nh; wl = nl;
      }
      // bias + activation + split of element g: acc index g <-> hidden unit (g/4)*8 + 4*lh + g%4 of the chunk
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const float u = fmaf(h[t][g], p.inv_s1, bv[g]);
        const float v = gelu_erf(u);              // nn.GELU (erf form)
        _Float16 f, l;
        split_f16(v, f, l);
        gh[t][g >> 3][g & 7] = f;
        gl[t][g >> 3][g & 7] = l;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      if (P1) h[t] = hn[t];
#pragma unroll
      for (int s = 0; s < 2; ++s) { hh[t][s] = gh[t][s]; hl[t][s] = gl[t][s]; }
    }
  };
  using yes = std::true_type;
  using no = std::false_type;

  // chunks 0 and 1's W1 have landed when only the ring prologue's NS-1 stages are outstanding
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * DPW) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  phase1(smem + (NS - 1) * SB + lane * 16, h);
  // iteration 0: GELU of chunk 0 beside phase 1 of chunk 1 (whose W1 sits in the W2 half of the prologue's slot: the body
  // addresses phase-1 fragments relative to a stage base, so hand it the base that puts them there)
  if (p.nch > 1) body(no(), yes(), 0, smem + (NS - 1) * SB + W1B + lane * 16);
  else body(no(), no(), 0, smem + lane * 16);

  // top of iteration j >= 1: stage j-1 has landed (mine: counted vmcnt, the NS-2 stages behind it stay in flight;
  // everybody's: barrier); the barrier also says every wave is done with what iteration j-1 read (stage j-2, or the
  // prologue's slot), which the next DMA overwrites
  auto top = [&](int j) {
    if (j + NS - 3 < p.nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * DPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (j + NS - 2 < p.nch) issue_stage(j + NS - 2);
  };
  int j = 1;
  for (; j + 1 < p.nch; ++j) {
    top(j);
    body(yes(), yes(), j, smem + ((j - 1) % NS) * SB + lane * 16);
  }
  if (j < p.nch) {      // j = nch - 1 (nch >= 2): no chunk j + 1
    top(j);
    body(yes(), no(), j, smem + ((j - 1) % NS) * SB + lane * 16);
    ++j;
  }
  top(j);               // j = nch: phase 2 of the last chunk
  {
    const unsigned char* w2 = smem + ((j - 1) % NS) * SB + lane * 16 + W1B;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const v8h wh = *reinterpret_cast<const v8h*>(w2 + ((s * CT + ct) * 2 + 0) * 1024);
        const v8h wl = *reinterpret_cast<const v8h*>(w2 + ((s * CT + ct) * 2 + 1) * 1024);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          if (!kSingleProduct) {
            o[t][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hh[t][s], o[t][ct], 0, 0, 0);
            o[t][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hl[t][s], o[t][ct], 0, 0, 0);
          }
          o[t][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hh[t][s], o[t][ct], 0, 0, 0);
        }
      }
  }

  // ---- epilogue: lane (li, lh) holds, for row li, channels ct*32 + q*8 + 4*lh + 0..3
  bool bad = false;
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const long row = row0 + t * 32 + li;
    const bool ok = row < p.M;
    const long rr = ok ? row : 0;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float4 rv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = ct * 32 + q * 8 + 4 * lh;
        rv[q] = p.res ? *reinterpret_cast<const float4*>(p.res + rr * p.ldr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = ct * 32 + q * 8 + 4 * lh;
        const float4 b = *reinterpret_cast<const float4*>(p.b2 + c);
        float4 v;
        v.x = fmaf(o[t][ct][q * 4 + 0], p.inv_s2, b.x) + rv[q].x;
        v.y = fmaf(o[t][ct][q * 4 + 1], p.inv_s2, b.y) + rv[q].y;
        v.z = fmaf(o[t][ct][q * 4 + 2], p.inv_s2, b.z) + rv[q].z;
        v.w = fmaf(o[t][ct][q * 4 + 3], p.inv_s2, b.w) + rv[q].w;
        bad |= nonfinite(v.x) | nonfinite(v.y) | nonfinite(v.z) | nonfinite(v.w);
        if (ok) *reinterpret_cast<float4*>(p.y + row * p.ldy + c) = v;
      }
    }
  }
  report_nonfinite(p.status, bad);
}

template <int C, int TM, int NS, int NWV = 4>
static int launch_mlp(const MlpArgs& a, hipStream_t s) {
  constexpr int BM = 32 * NWV * TM;
  const long nb = (a.M + BM - 1) / BM;
  hipLaunchKernelGGL((mlp_fused_kernel<C, TM, NS, NWV>), dim3((unsigned)nb), dim3(64 * NWV), 0, s, a);
  return 0;
}

}  // namespace mspi

using namespace mspi;

extern "C" size_t mspi_mlp_packed_bytes(int C, int hidden) {
  return (size_t)(hidden / 32) * ((size_t)(C / 16) * 2048 + 2 * (size_t)(C / 32) * 2048);
}

extern "C" int mspi_mlp_fwd(const MspiMlpDesc* d, const void* x, const void* gamma, const void* beta, const void* w_packed,
                            const void* b1, const void* b2, const void* res, void* y, void* stream) {
  MSPI_REQUIRE(d && x && w_packed && b1 && b2 && y, "mspi_mlp_fwd: null argument");
  MSPI_REQUIRE(d->C == 96 || d->C == 192, "mspi_mlp_fwd: C = %d not supported (96, 192)", d->C);
  MSPI_REQUIRE(d->hidden % 32 == 0 && d->hidden >= 32 && d->hidden <= 1024, "mspi_mlp_fwd: hidden = %d", d->hidden);
  MSPI_REQUIRE(d->M >= 1 && d->M < (1L << 31), "mspi_mlp_fwd: M = %ld", (long)d->M);
  MSPI_REQUIRE(d->ldx >= d->C && d->ldy >= d->C && d->ldx % 4 == 0 && d->ldy % 4 == 0 && (!res || (d->ldr >= d->C && d->ldr % 4 == 0)),
               "mspi_mlp_fwd: row strides must be >= C and multiples of 4 floats");
  MSPI_REQUIRE(d->act == MSPI_ACT_GELU, "mspi_mlp_fwd: only MSPI_ACT_GELU between the layers (act = %d)", d->act);
  MSPI_REQUIRE(!d->ln || (gamma && beta), "mspi_mlp_fwd: LayerNorm needs gamma and beta");
  MSPI_REQUIRE(d->w1_scale > 0.f && d->w2_scale > 0.f, "mspi_mlp_fwd: weight scales must be positive");
  MlpArgs a;
  a.x = (const float*)x; a.gamma = (const float*)gamma; a.beta = (const float*)beta; a.wp = (const unsigned char*)w_packed;
  a.b1 = (const float*)b1; a.b2 = (const float*)b2; a.res = (const float*)res; a.y = (float*)y;
  a.M = d->M; a.ldx = d->ldx; a.ldr = d->ldr; a.ldy = d->ldy;
  a.nch = d->hidden / 32; a.ln = d->ln; a.act = d->act; a.eps = d->eps;
  a.inv_s1 = 1.0f / d->w1_scale; a.inv_s2 = 1.0f / d->w2_scale;
  a.status = g_status_word;
  static const int variant = getenv("MSPI_MLP_TM") ? atoi(getenv("MSPI_MLP_TM")) : 0;
  int rc;
  MSPI_REQUIRE(d->C != 96 || d->hidden <= 512, "mspi_mlp_fwd: hidden = %d > 512 with C = 96", d->hidden);
  if (d->C == 96) rc = (variant == 2) ? launch_mlp<96, 2, 4>(a, (hipStream_t)stream) :
                       launch_mlp<96, 1, 3>(a, (hipStream_t)stream);
  // C = 192: 8 waves (256 rows) per workgroup where that still gives every CU a workgroup (measured at M = 100352: 315.9 ->
  // 277.6 us); MSPI_MLP_TM=4 / 8 force one form for an A/B
  else if (variant == 8 || (variant != 4 && a.M >= 256L * 256)) rc = launch_mlp<192, 1, 3, 8>(a, (hipStream_t)stream);
  else rc = launch_mlp<192, 1, 3>(a, (hipStream_t)stream);
  (void)rc;
  return check_launch("mspi_mlp_fwd");
}

// =====================================================================================================
// Row-stationary thin GEMM:  y[M, N] = act( x[M, K] . W^T + bias (+ res) )  for K <= 224, N small (X3D's 1x1x1 layers).
//
// The tiled implicit GEMM is the wrong tool for these layers: K = 24..216 is 1-7 pipeline steps, so a workgroup is
// prologue + epilogue and the kernel is one exposed memory latency after another (17-25 us for 30 MB of traffic,
// profiles/r01).  Here -- phase 1 of the fused MLP above, with a store epilogue -- a wave loads its 32 rows ONCE
// (optionally x <- swish(x * gate[sample]): the squeeze-excite prologue of X3D's `c` conv), keeps them as f16 hi/lo MFMA
// B-operand fragments, and walks the output columns in chunks of 32:  Y^T chunk = W[chunk rows][:] . X^T.  All weights
// (fragment order, engine.pack_rowgemm) are brought into LDS by one burst of LDS-DMA at kernel start, so after the
// single load phase the kernel only computes and stores.
namespace mspi {

struct RowGemmArgs {
  const float* x; const unsigned char* wp; const float* bias; const float* res; const float* gate; float* y;
  long M, ldx, ldr, ldy, ldg;
  int K, N;           // storage columns of x / y (multiples of 4)
  int nch;            // ceil(N / 32)
  int cps;            // chunks per workgroup: grid.y workgroups share a row tile and split the output columns
  int act;
  float inv_s;
  int rows_per_sample;
  int* status;
};

constexpr int RG_LDS = 64 * 1024;   // dynamic LDS budget of one workgroup (rowgemm_cps keeps cps * (SB + 128) below it: no opt-in needed)

template <int KSB, bool GATE>
__global__ __launch_bounds__(256, 2) void rowgemm_kernel(const RowGemmArgs p) {
  constexpr int SB = KSB * 2048;                 // bytes of one 32-column chunk: [ks][hi,lo][lane][8 halves]
  extern __shared__ __attribute__((aligned(16))) unsigned char rg_smem[];   // cps * SB weights, then cps*32 bias floats
  const int j0 = blockIdx.y * p.cps;
  const int nj = min(p.cps, p.nch - j0);
  float* bs = reinterpret_cast<float*>(rg_smem + (size_t)p.cps * SB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long row = (long)blockIdx.x * 128 + wave * 32 + li;
  const bool rok = row < p.M;
  const long rr = rok ? row : p.M - 1;

  // weights of this workgroup's chunks: nj*SB/1024 pieces of 1 KB, dealt round-robin to the 4 waves
  const int pieces = nj * (SB / 1024);
  for (int i = wave; i < pieces; i += 4)
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(p.wp + (long)j0 * SB + (long)i * 1024 + lane * 16),
                                     (lds_void*)(rg_smem + (long)i * 1024), 16, 0, 0);
  // rows
  float xr[KSB][8];
  const float* xp = p.x + rr * p.ldx + 8 * lh;
#pragma unroll
  for (int ks = 0; ks < KSB; ++ks) {
    const int k = 16 * ks + 8 * lh;
    const bool k0 = k < p.K, k1 = k + 4 < p.K;
    const float4 a = *reinterpret_cast<const float4*>(xp + (k0 ? 16 * ks : -8 * lh));
    const float4 b = *reinterpret_cast<const float4*>(xp + (k1 ? 16 * ks + 4 : -8 * lh));
    xr[ks][0] = k0 ? a.x : 0.f; xr[ks][1] = k0 ? a.y : 0.f; xr[ks][2] = k0 ? a.z : 0.f; xr[ks][3] = k0 ? a.w : 0.f;
    xr[ks][4] = k1 ? b.x : 0.f; xr[ks][5] = k1 ? b.y : 0.f; xr[ks][6] = k1 ? b.z : 0.f; xr[ks][7] = k1 ? b.w : 0.f;
  }
  if (GATE) {
    const float* gp = p.gate + (long)((unsigned)rr / (unsigned)p.rows_per_sample) * p.ldg + 8 * lh;   // 32-bit division (M < 2^31)
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks) {
      const int k = 16 * ks + 8 * lh;
      const bool k0 = k < p.K, k1 = k + 4 < p.K;
      const float4 a = *reinterpret_cast<const float4*>(gp + (k0 ? 16 * ks : -8 * lh));
      const float4 b = *reinterpret_cast<const float4*>(gp + (k1 ? 16 * ks + 4 : -8 * lh));
      const float g8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = xr[ks][e] * g8[e];
        xr[ks][e] = fast_swish(v);     // Swish (zero stays zero: masked k contribute nothing)
      }
    }
  }
  for (int i = tid; i < nj * 32; i += 256) bs[i] = (p.bias && j0 * 32 + i < p.N) ? p.bias[j0 * 32 + i] : 0.f;
  v8h xh[KSB], xl[KSB];
#pragma unroll
  for (int ks = 0; ks < KSB; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 h, l;
      split_f16(xr[ks][e], h, l);
      xh[ks][e] = h; xl[ks][e] = l;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  bool bad = false;
  for (int jl = 0; jl < nj; ++jl) {
    const int j = j0 + jl;
    const unsigned char* st = rg_smem + (long)jl * SB + lane * 16;
    v16f h;
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks) {
      const v8h wh = *reinterpret_cast<const v8h*>(st + (ks * 2 + 0) * 1024);
      const v8h wl = *reinterpret_cast<const v8h*>(st + (ks * 2 + 1) * 1024);
      if (!kSingleProduct) {
        h = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[ks], h, 0, 0, 0);
        h = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[ks], h, 0, 0, 0);
      }
      h = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[ks], h, 0, 0, 0);
    }
    // lane (li, lh) holds, for row li, columns j*32 + q*8 + 4*lh + 0..3
    float4 rv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = j * 32 + q * 8 + 4 * lh;
      rv[q] = (p.res && c < p.N) ? *reinterpret_cast<const float4*>(p.res + rr * p.ldr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = j * 32 + q * 8 + 4 * lh;
      const float4 b = *reinterpret_cast<const float4*>(bs + c - j0 * 32);
      float4 v;
      v.x = fmaf(h[q * 4 + 0], p.inv_s, b.x) + rv[q].x;
      v.y = fmaf(h[q * 4 + 1], p.inv_s, b.y) + rv[q].y;
      v.z = fmaf(h[q * 4 + 2], p.inv_s, b.z) + rv[q].z;
      v.w = fmaf(h[q * 4 + 3], p.inv_s, b.w) + rv[q].w;
      bad |= nonfinite(v.x) | nonfinite(v.y) | nonfinite(v.z) | nonfinite(v.w);
      if (p.act == MSPI_ACT_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      } else if (p.act != MSPI_ACT_NONE) {
        v.x = act_apply(v.x, p.act); v.y = act_apply(v.y, p.act); v.z = act_apply(v.z, p.act); v.w = act_apply(v.w, p.act);
      }
      if (rok && c < p.N) *reinterpret_cast<float4*>(p.y + row * p.ldy + c) = v;
    }
  }
  report_nonfinite(p.status, bad);
}

template <int KSB>
static int launch_rowgemm(const RowGemmArgs& a, size_t lds, hipStream_t s) {
  const dim3 grid((unsigned)((a.M + 127) / 128), (unsigned)((a.nch + a.cps - 1) / a.cps));
  if (a.gate) hipLaunchKernelGGL((rowgemm_kernel<KSB, true>), grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL((rowgemm_kernel<KSB, false>), grid, dim3(256), lds, s, a);
  return 0;
}

}  // namespace mspi

static int rowgemm_ksb(int K) { return K <= 32 ? 2 : K <= 64 ? 4 : K <= 128 ? 8 : K <= 224 ? 14 : 0; }

extern "C" size_t mspi_rowgemm_packed_bytes(int32_t K, int32_t N) {
  const int ksb = rowgemm_ksb(K);
  return ksb ? (size_t)((N + 31) / 32) * ksb * 2048 : 0;
}

// chunks per workgroup: all of them when the row tiles alone fill the chip; otherwise the output columns are split over
// grid.y so that (a) the grid reaches ~2 workgroups per CU and (b) a workgroup's weights fit in LDS with room for a neighbour
static int rowgemm_cps(long M, int nch, int ksb) {
  const long row_tiles = (M + 127) / 128;
  int split = 1;
  if (row_tiles < 512) split = (int)((512 + row_tiles - 1) / row_tiles);
  if (split > nch) split = nch;
  int cps = (nch + split - 1) / split;
  const int cap = RG_LDS / (ksb * 2048 + 128);        // the kernel's static LDS array
  if (cps > cap) cps = cap;
  return cps < 1 ? 1 : cps;
}

extern "C" int mspi_rowgemm_supported(int32_t K, int32_t N) {
  return rowgemm_ksb(K) != 0 && N >= 4 && N <= 1024;
}

extern "C" int mspi_rowgemm_fwd(const MspiRowGemmDesc* d, const void* x, const void* w_packed, const void* bias, const void* res,
                                const void* gate, void* y, void* stream) {
  MSPI_REQUIRE(d && x && w_packed && y, "mspi_rowgemm_fwd: null argument");
  MSPI_REQUIRE(d->M >= 1 && d->M < (1L << 31) && d->K >= 4 && d->N >= 4 && d->K % 4 == 0 && d->N % 4 == 0,
               "mspi_rowgemm_fwd: M = %ld, K = %d, N = %d (K, N: storage columns, multiples of 4)", (long)d->M, d->K, d->N);
  MSPI_REQUIRE(mspi_rowgemm_supported(d->K, d->N), "mspi_rowgemm_fwd: K = %d / N = %d outside the thin-GEMM range", d->K, d->N);
  MSPI_REQUIRE(d->ldx >= d->K && d->ldy >= d->N && d->ldx % 4 == 0 && d->ldy % 4 == 0 && (!res || (d->ldr >= d->N && d->ldr % 4 == 0)),
               "mspi_rowgemm_fwd: row strides must cover the row and be multiples of 4 floats");
  MSPI_REQUIRE(!gate || (d->rows_per_sample > 0 && d->ldg >= d->K && d->ldg % 4 == 0), "mspi_rowgemm_fwd: gate needs rows_per_sample and ldg");
  MSPI_REQUIRE(d->w_scale > 0.f, "mspi_rowgemm_fwd: w_scale must be positive");
  RowGemmArgs a;
  a.x = (const float*)x; a.wp = (const unsigned char*)w_packed; a.bias = (const float*)bias; a.res = (const float*)res;
  a.gate = (const float*)gate; a.y = (float*)y;
  a.M = d->M; a.ldx = d->ldx; a.ldr = d->ldr; a.ldy = d->ldy; a.ldg = d->ldg;
  a.K = d->K; a.N = d->N; a.nch = (d->N + 31) / 32; a.act = d->act; a.inv_s = 1.0f / d->w_scale;
  a.rows_per_sample = d->rows_per_sample;
  a.status = g_status_word;
  const int ksb = rowgemm_ksb(d->K);
  a.cps = rowgemm_cps(d->M, a.nch, ksb);
  const size_t lds = (size_t)a.cps * (ksb * 2048 + 128);
  int rc;
  switch (ksb) {
    case 2: rc = launch_rowgemm<2>(a, lds, (hipStream_t)stream); break;
    case 4: rc = launch_rowgemm<4>(a, lds, (hipStream_t)stream); break;
    case 8: rc = launch_rowgemm<8>(a, lds, (hipStream_t)stream); break;
    default: rc = launch_rowgemm<14>(a, lds, (hipStream_t)stream); break;
  }
  MSPI_REQUIRE(rc == 0, "mspi_rowgemm_fwd: could not reserve %zu bytes of LDS", lds);
  return check_launch("mspi_rowgemm_fwd");
}

// =====================================================================================================
// X3D block seam:  y = relu( Wc . g(u) + bc + x )          (this block's `c` conv, residual, ReLU  -> the block output)
//                  t = relu( Wa'. y + ba' )                (the NEXT block's `a` conv)
// with g(u) = swish(u * gate[sample]) for squeeze-excite blocks, identity otherwise.
//
// Stages 3-4 of X3D-L are 9 / 24 stride-1 blocks of three small layers each; as separate launches `c` (21-26 us) and the
// following `a` (17 us) are each one exposed load -> compute -> store round trip for a few MB (profiles/r03_x3d_stage_kernel.txt).
// This is the fused MLP above with the roles changed: the wave keeps its 32 rows of u as MFMA B fragments, walks the block
// width (y's columns) in chunks of 32 -- phase 1 computes the chunk of y^T, its epilogue adds bias + residual, applies the
// ReLU, STORES the chunk of y and keeps it (split) as phase 2's B operand -- and accumulates all of t^T across the chunks.
// Packing = mspi_mlp's with C = the dw width padded to 128 / 224 and hidden = the block width padded to 32
// (engine.pack_x3d_ca); padded channels are zero weights / zero bias and are never loaded or stored.
namespace mspi {

struct CaArgs {
  const float* u; const float* gate; const unsigned char* wp; const float* b1; const float* b2; const float* res;
  float* y; float* t;
  long M, ldu, ldg, ldr, ldy, ldt;
  int D, Cx;          // stored columns of u / t and of x / y (multiples of 4)
  int nch;            // ceil(Cx / 32)
  int rows_per_sample;
  float inv_s1, inv_s2;
  int* status;
};

template <int C, bool GATE>
__global__ __launch_bounds__(256, 1) void x3d_ca_kernel(const CaArgs p) {
  constexpr int NS = 2;
  constexpr int KS = C / 16, CT = C / 32;
  constexpr int W1B = KS * 2048, W2B = 2 * CT * 2048, SB = W1B + W2B;
  constexpr int DPW = SB / 4096;
  static_assert(C % 32 == 0 && SB % 4096 == 0, "C: multiple of 32");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * SB + 1024];   // the ring, then bc (<= 256 floats)
  float* b1s = reinterpret_cast<float*>(smem + NS * SB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long row = (long)blockIdx.x * 128 + wave * 32 + li;
  const bool rok = row < p.M;
  const long rr = rok ? row : p.M - 1;

  auto issue_stage = [&](int j) {
    const unsigned char* src = p.wp + (long)j * SB + (wave * DPW) * 1024 + lane * 16;
    unsigned char* dst = smem + (j % NS) * SB + (wave * DPW) * 1024;
#pragma unroll
    for (int d = 0; d < DPW; ++d)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + d * 1024), (lds_void*)(dst + d * 1024), 16, 0, 0);
  };
  for (int j = 0; j < NS - 1 && j < p.nch; ++j) issue_stage(j);

  float xr[KS][8];
  const float* xp = p.u + rr * p.ldu + 8 * lh;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k = 16 * ks + 8 * lh;
    const bool k0 = k < p.D, k1 = k + 4 < p.D;
    const float4 a = *reinterpret_cast<const float4*>(xp + (k0 ? 16 * ks : -8 * lh));
    const float4 b = *reinterpret_cast<const float4*>(xp + (k1 ? 16 * ks + 4 : -8 * lh));
    xr[ks][0] = k0 ? a.x : 0.f; xr[ks][1] = k0 ? a.y : 0.f; xr[ks][2] = k0 ? a.z : 0.f; xr[ks][3] = k0 ? a.w : 0.f;
    xr[ks][4] = k1 ? b.x : 0.f; xr[ks][5] = k1 ? b.y : 0.f; xr[ks][6] = k1 ? b.z : 0.f; xr[ks][7] = k1 ? b.w : 0.f;
  }
  if (GATE) {
    const float* gp = p.gate + (long)((unsigned)rr / (unsigned)p.rows_per_sample) * p.ldg + 8 * lh;   // 32-bit division (M < 2^31)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 16 * ks + 8 * lh;
      const bool k0 = k < p.D, k1 = k + 4 < p.D;
      const float4 a = *reinterpret_cast<const float4*>(gp + (k0 ? 16 * ks : -8 * lh));
      const float4 b = *reinterpret_cast<const float4*>(gp + (k1 ? 16 * ks + 4 : -8 * lh));
      const float g8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = xr[ks][e] * g8[e];
        xr[ks][e] = fast_swish(v);     // Swish (zero stays zero: masked k contribute nothing)
      }
    }
  }
  for (int i = tid; i < p.nch * 32; i += 256) b1s[i] = i < p.Cx ? p.b1[i] : 0.f;
  v8h xh[KS], xl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 h, l;
      split_f16(xr[ks][e], h, l);
      xh[ks][e] = h; xl[ks][e] = l;
    }

  v16f o[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[ct][i] = 0.f;

  bool bad = false;
  for (int j = 0; j < p.nch; ++j) {
    // stage j has landed (vmcnt(0): also this wave's y stores of chunk j-1); the barrier also says every wave is done
    // reading stage j-1, whose slot the next DMA overwrites
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (j + NS - 1 < p.nch) issue_stage(j + NS - 1);
    // residual chunk: in flight under phase 1
    float4 rv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = j * 32 + q * 8 + 4 * lh;
      rv[q] = c < p.Cx ? *reinterpret_cast<const float4*>(p.res + rr * p.ldr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const unsigned char* st = smem + (j % NS) * SB + lane * 16;
    v16f h;
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const v8h wh = *reinterpret_cast<const v8h*>(st + (ks * 2 + 0) * 1024);
      const v8h wl = *reinterpret_cast<const v8h*>(st + (ks * 2 + 1) * 1024);
      if (!kSingleProduct) {
        h = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[ks], h, 0, 0, 0);
        h = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[ks], h, 0, 0, 0);
      }
      h = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[ks], h, 0, 0, 0);
    }
    // ---- y chunk: acc index i <-> column j*32 + (i/4)*8 + 4*lh + i%4;  bias + residual + ReLU, store, split
    v8h hh[2], hl[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = j * 32 + q * 8 + 4 * lh;
      const float4 b = *reinterpret_cast<const float4*>(b1s + c);
      float4 v;
      v.x = fmaf(h[q * 4 + 0], p.inv_s1, b.x) + rv[q].x;
      v.y = fmaf(h[q * 4 + 1], p.inv_s1, b.y) + rv[q].y;
      v.z = fmaf(h[q * 4 + 2], p.inv_s1, b.z) + rv[q].z;
      v.w = fmaf(h[q * 4 + 3], p.inv_s1, b.w) + rv[q].w;
      bad |= rok && (nonfinite(v.x) | nonfinite(v.y) | nonfinite(v.z) | nonfinite(v.w));
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      if (rok && c < p.Cx) *reinterpret_cast<float4*>(p.y + row * p.ldy + c) = v;
      const float v4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        _Float16 f, l;
        split_f16(v4[r], f, l);
        hh[q >> 1][(q & 1) * 4 + r] = f;
        hl[q >> 1][(q & 1) * 4 + r] = l;
      }
    }
    // ---- phase 2: t^T += Wa'[:, chunk] . y chunk
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const v8h wh = *reinterpret_cast<const v8h*>(st + W1B + ((s * CT + ct) * 2 + 0) * 1024);
        const v8h wl = *reinterpret_cast<const v8h*>(st + W1B + ((s * CT + ct) * 2 + 1) * 1024);
        if (!kSingleProduct) {
          o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hh[s], o[ct], 0, 0, 0);
          o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hl[s], o[ct], 0, 0, 0);
        }
        o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hh[s], o[ct], 0, 0, 0);
      }
  }

  // ---- t: lane (li, lh) holds, for row li, channels ct*32 + q*8 + 4*lh + 0..3
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = ct * 32 + q * 8 + 4 * lh;
      if (c < p.D) {
        const float4 b = *reinterpret_cast<const float4*>(p.b2 + c);
        float4 v;
        v.x = fmaf(o[ct][q * 4 + 0], p.inv_s2, b.x);
        v.y = fmaf(o[ct][q * 4 + 1], p.inv_s2, b.y);
        v.z = fmaf(o[ct][q * 4 + 2], p.inv_s2, b.z);
        v.w = fmaf(o[ct][q * 4 + 3], p.inv_s2, b.w);
        bad |= rok && (nonfinite(v.x) | nonfinite(v.y) | nonfinite(v.z) | nonfinite(v.w));
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (rok) *reinterpret_cast<float4*>(p.t + row * p.ldt + c) = v;
      }
    }
  report_nonfinite(p.status, bad);
}

}  // namespace mspi

static int x3d_ca_c(int D) { return D <= 128 ? 128 : D <= 224 ? 224 : 0; }

extern "C" int mspi_x3d_ca_supported(int32_t D, int32_t Cx) {
  return x3d_ca_c(D) != 0 && D >= 4 && D % 4 == 0 && Cx >= 4 && Cx % 4 == 0 && Cx <= 256;
}

extern "C" size_t mspi_x3d_ca_packed_bytes(int32_t D, int32_t Cx) {
  const int c = x3d_ca_c(D);
  return c ? mspi_mlp_packed_bytes(c, (Cx + 31) / 32 * 32) : 0;
}

extern "C" int mspi_x3d_ca_fwd(const MspiX3dCaDesc* d, const void* u, const void* gate, const void* w_packed, const void* bc,
                               const void* ba, const void* res, void* y, void* t, void* stream) {
  MSPI_REQUIRE(d && u && w_packed && bc && ba && res && y && t, "mspi_x3d_ca_fwd: null argument");
  MSPI_REQUIRE(mspi_x3d_ca_supported(d->D, d->Cx), "mspi_x3d_ca_fwd: D = %d / Cx = %d outside the kernel's range", d->D, d->Cx);
  MSPI_REQUIRE(d->M >= 1 && d->M < (1L << 31), "mspi_x3d_ca_fwd: M = %ld", (long)d->M);
  MSPI_REQUIRE(d->ldu >= d->D && d->ldt >= d->D && d->ldr >= d->Cx && d->ldy >= d->Cx &&
               d->ldu % 4 == 0 && d->ldt % 4 == 0 && d->ldr % 4 == 0 && d->ldy % 4 == 0,
               "mspi_x3d_ca_fwd: row strides must cover the row and be multiples of 4 floats");
  MSPI_REQUIRE(!gate || (d->rows_per_sample > 0 && d->ldg >= d->D && d->ldg % 4 == 0), "mspi_x3d_ca_fwd: gate needs rows_per_sample and ldg");
  MSPI_REQUIRE(d->wc_scale > 0.f && d->wa_scale > 0.f, "mspi_x3d_ca_fwd: weight scales must be positive");
  CaArgs a;
  a.u = (const float*)u; a.gate = (const float*)gate; a.wp = (const unsigned char*)w_packed; a.b1 = (const float*)bc;
  a.b2 = (const float*)ba; a.res = (const float*)res; a.y = (float*)y; a.t = (float*)t;
  a.M = d->M; a.ldu = d->ldu; a.ldg = d->ldg; a.ldr = d->ldr; a.ldy = d->ldy; a.ldt = d->ldt;
  a.D = d->D; a.Cx = d->Cx; a.nch = (d->Cx + 31) / 32; a.rows_per_sample = d->rows_per_sample;
  a.inv_s1 = 1.0f / d->wc_scale; a.inv_s2 = 1.0f / d->wa_scale;
  a.status = g_status_word;
  const dim3 grid((unsigned)((a.M + 127) / 128));
  hipStream_t s = (hipStream_t)stream;
  if (x3d_ca_c(d->D) == 128) {
    if (gate) hipLaunchKernelGGL((x3d_ca_kernel<128, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((x3d_ca_kernel<128, false>), grid, dim3(256), 0, s, a);
  } else {
    if (gate) hipLaunchKernelGGL((x3d_ca_kernel<224, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((x3d_ca_kernel<224, false>), grid, dim3(256), 0, s, a);
  }
  return check_launch("mspi_x3d_ca_fwd");
}
