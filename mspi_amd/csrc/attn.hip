// Fused multi-head attention, fp32 on v_mfma_f32_32x32x2_f32, online softmax.
//
// Layout trick (CDNA accumulator-as-operand): each wave owns 32 queries and computes the
// TRANSPOSED score tile  S^T[key][q] = K . Q^T  so that the query sits on the lane
// (column) and the 32 keys of the tile sit in the 16 accumulator registers x 2 lane
// halves.  Then
//   * softmax statistics per query are reductions over a lane's own registers plus one
//     cross-half shuffle -- no LDS, no row shuffles;
//   * the probabilities are already the B operand of  O^T[d][q] += V^T[d][key] . P^T[key][q]:
//     MFMA step r consumes register r of P directly (lane half h supplies key
//     (r&3)+8(r>>2)+4h, and the A operand V^T is read from the LDS V tile at that key row),
//     so P never moves between lanes or through memory.
// K and V tiles (32 keys) are staged once per workgroup (4 waves = 128 queries) in LDS
// rows padded by 4 floats: the K fragment reads are conflict-free ds_read_b128, the V
// fragment reads conflict-free ds_read_b32.
#include "common.h"

namespace mspi {

typedef float v16f __attribute__((ext_vector_type(16)));

struct AttnArgs {
  const float* q;
  const float* k;
  const float* v;
  const float* res;    // optional, added to the output (o strides)
  const float* biasT;  // optional [Hh][Nk][Nq]: transposed additive bias (key-major so a lane row is coalesced)
  const float* maskT;  // optional [nmask][Nk][Nq]: additive mask, sequence b uses slice b % nmask
  const int* tok_idx;  // optional [nwin][N]: token t of sequence (sample, win) lives at row tok_idx[win][t] of the sample
  float* o;
  int B, Hh, Nq, Nk, nmask, nwin;
  long q_sB, q_sH, q_sT, k_sB, k_sH, k_sT, v_sB, v_sH, v_sT, o_sB, o_sH, o_sT;
  float scale;
  // rounded up to 32, already scaled by KSC / VSC, zero beyond Nk
  _Float16* kp;
  _Float16* vp;
  int Nkp;
  // key split (gridDim.z > 1): slice z of the key tiles leaves its unnormalised O^T, running maximum and running sum here
  float* part_o;       // [z][B*Hh][Nq][DV]
  float* part_ml;      // [z][B*Hh][Nq][2]
};

// D = head dim of Q/K (the contraction of S), DV = head dim of V / O.  They differ for MViT, whose decomposed
// relative-position terms ride along as extra Q/K columns (mvit_aug_kernel below).
template <int D, int DV>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs p) {
  constexpr int LDD = D + 4;
  constexpr int LDV = DV + 4;
  constexpr int NC = D / 8;    // float4 chunks of one lane half
  constexpr int NT = DV / 32;  // 32-wide output tiles along d
  __shared__ __attribute__((aligned(16))) float smem[32 * LDD + 32 * LDV];
  float* Ks = smem;
  float* Vs = smem + 32 * LDD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y / p.Hh, h = blockIdx.y % p.Hh;
  const int q = blockIdx.x * 128 + wave * 32 + li;
  const bool qok = q < p.Nq;
  // windowed sequences (Swin): b = sample * nwin + win; rows are looked up, so the cyclic shift, the window
  // partition and their inverses are pure index arithmetic -- no gather / scatter pass over the activations
  const int sample = p.tok_idx ? b / p.nwin : b;
  const int* tix = p.tok_idx ? p.tok_idx + (long)(b % p.nwin) * p.Nk : nullptr;
  const int qrow = qok ? (tix ? tix[q] : q) : 0;

  float4 qr[NC];
  {
    const float* qp = p.q + (long)sample * p.q_sB + (long)h * p.q_sH + (long)qrow * p.q_sT;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qok) t = *reinterpret_cast<const float4*>(qp + 4 * (2 * j + lh));
      qr[j] = make_float4(t.x * p.scale, t.y * p.scale, t.z * p.scale, t.w * p.scale);
    }
  }

  v16f acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const float* kb = p.k + (long)sample * p.k_sB + (long)h * p.k_sH;
  const float* vb = p.v + (long)sample * p.v_sB + (long)h * p.v_sH;

  for (int k0 = 0; k0 < p.Nk; k0 += 32) {
    __syncthreads();  // previous tile fully consumed
    for (int idx = tid; idx < 32 * (D / 4); idx += 256) {
      const int row = idx / (D / 4), c4 = idx - row * (D / 4);
      const bool ok = k0 + row < p.Nk;
      const int kr = ok ? (tix ? tix[k0 + row] : k0 + row) : 0;
      const float4 kv = *reinterpret_cast<const float4*>(kb + (ok ? (long)kr * p.k_sT + c4 * 4 : 0));
      *reinterpret_cast<float4*>(&Ks[row * LDD + c4 * 4]) = ok ? kv : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int idx = tid; idx < 32 * (DV / 4); idx += 256) {
      const int row = idx / (DV / 4), c4 = idx - row * (DV / 4);
      const bool ok = k0 + row < p.Nk;
      const int kr = ok ? (tix ? tix[k0 + row] : k0 + row) : 0;
      const float4 vv = *reinterpret_cast<const float4*>(vb + (ok ? (long)kr * p.v_sT + c4 * 4 : 0));
      *reinterpret_cast<float4*>(&Vs[row * LDV + c4 * 4]) = ok ? vv : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    // S^T = K . Q^T  (A = K rows on the lane, B = Q^T from registers)
    v16f s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const float4 kf = *reinterpret_cast<const float4*>(&Ks[li * LDD + 4 * (2 * j + lh)]);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qr[j].x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qr[j].y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qr[j].z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qr[j].w, s, 0, 0, 0);
    }

    // online softmax over the 32 keys of this tile (16 in my registers, 16 in lane^32's)
    float mt = -INFINITY;
    if (p.biasT || p.maskT) {   // wave-uniform: Swin's learned bias table / shifted-window mask
      const float* bt = p.biasT ? p.biasT + (long)h * p.Nk * p.Nq : nullptr;
      const float* mk = p.maskT ? p.maskT + (long)(b % p.nmask) * p.Nk * p.Nq : nullptr;
      float add[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const long o = (qok && key < p.Nk) ? (long)key * p.Nq + q : 0;
        add[r] = (bt ? bt[o] : 0.f) + (mk ? mk[o] : 0.f);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] += add[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (key >= p.Nk) s[r] = -INFINITY;
      mt = fmaxf(mt, s[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);  // finite: every tile holds at least one valid key
    const float alpha = __expf(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __expf(s[r] - m_new);
      ps += s[r];
    }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;

    // O^T += V^T . P^T  (A = V^T read from LDS at the key row my half supplies, B = P registers)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow = (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float vf = Vs[krow * LDV + t * 32 + li];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], acc[t], 0, 0, 0);
      }
    }
  }

  if (qok) {
    const float inv = 1.f / l_run;
    const long oo = (long)sample * p.o_sB + (long)h * p.o_sH + (long)qrow * p.o_sT;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // registers 4g..4g+3 are 4 consecutive d: d = t*32 + 8g + 4*lh + (0..3)
        float4 o4 = make_float4(acc[t][4 * g] * inv, acc[t][4 * g + 1] * inv, acc[t][4 * g + 2] * inv,
                                acc[t][4 * g + 3] * inv);
        const int dd = t * 32 + 8 * g + 4 * lh;
        if (p.res) {
          const float4 rr = *reinterpret_cast<const float4*>(p.res + oo + dd);
          o4.x += rr.x; o4.y += rr.y; o4.z += rr.z; o4.w += rr.w;
        }
        *reinterpret_cast<float4*>(p.o + oo + dd) = o4;
      }
  }
}

// ------------------------------------------------------------------ the same attention on the f16 matrix pipe
// f16x3 split products (see conv_common.h): every operand x = hi + lo in f16, x.y = hi.hi + hi.lo + lo.hi accumulated
// in fp32 by v_mfma_f32_32x32x16_f16 -- fp32-accurate, 48 MFMA x 32 cycles per (32 queries x 32 keys x 128 d) tile instead
// of 128 x 64 cycles on the fp32 pipe.  Differences to attn_kernel above:
//   * K is staged as two f16 planes [key][D] (rows padded by 8 halves: conflict-free ds_read_b128), Q is split once into
//     registers as the B operand of S^T = K . Q^T;
//   * V is staged TRANSPOSED as two f16 planes Vt[d][key] (36-half rows): the A operand of O^T += V^T . P^T needs, per
//     lane (d, half), 8 keys of one column d.  The key <-> k-slot order is free as long as both operands agree, so it is
//     chosen to be the order in which the S^T accumulator already holds the probabilities:
//         k-slot (step s, half h, e)  <->  key 16 s + 8 (e >> 2) + 4 h + (e & 3)
//     i.e. P needs no movement at all (registers 8s .. 8s+7 of the accumulator, split hi/lo in place), and the V^T
//     fragment is two 8-byte reads of row d (keys 16s+4h .. +3 and 16s+8+4h .. +3).
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void split4(const float4 v, v4h& hi, v4h& lo) {
  const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    _Float16 h, l;
    split_f16(a[e], h, l);
    hi[e] = h; lo[e] = l;
  }
}

#ifndef MSPI_ATT_PV_DROP
#define MSPI_ATT_PV_DROP 0
#endif
constexpr float ATT_KSC = 16.f, ATT_VSC = 16.f, ATT_PSC = 1024.f;

// K and V of every (sequence, head) split ONCE into the f16 hi/lo planes the attention kernel stages: every query tile of
// that head (196 workgroups at Nq = 25088) used to redo this split on its own copy -- ~200 VALU instructions per thread and
// 32-key step, beside ~60 MFMAs.  One workgroup per 32-key tile; same arithmetic as the in-kernel staging, so the planes
// hold bit for bit what attn_f16x3_kernel<.., false> puts into LDS.
// IMG: the planes are written as per-tile LDS IMAGES -- for each (sequence, head) and 32-key tile the K tile [hi, lo][32][D + 8]
// and the V^T tile [hi, lo][DV][36] exactly as attn_pipe_kernel keeps them in LDS (row pads included, the V image rounded up to
// a multiple of 1 KB), each one contiguous, so that a tile is staged by plain 1-KB LDS-DMA pieces.
template <int D, int DV> struct AttnImg {
  static constexpr int KP = D + 8, VP = 36;
  static constexpr int KTI = 2 * 32 * KP;                               // halves per K image (a multiple of 512: whole KB)
  static constexpr int VTI = (2 * DV * VP * 2 + 1023) / 1024 * 512;     // halves per V image, rounded up to whole KB
};

template <int D, int DV, bool IMG = false>
__global__ __launch_bounds__(256) void attn_kv_planes_kernel(const AttnArgs p) {
  const int tid = threadIdx.x;
  const int b = blockIdx.y / p.Hh, h = blockIdx.y % p.Hh;
  const int k0 = blockIdx.x * 32;
  const int sample = p.tok_idx ? b / p.nwin : b;
  const int* tix = p.tok_idx ? p.tok_idx + (long)(b % p.nwin) * p.Nk : nullptr;
  const float* kb = p.k + (long)sample * p.k_sB + (long)h * p.k_sH;
  const float* vb = p.v + (long)sample * p.v_sB + (long)h * p.v_sH;
  typedef AttnImg<D, DV> I;
  const long tile = (long)blockIdx.y * (p.Nkp >> 5) + blockIdx.x;
  // plain layout: planes [hi, lo][Nkp][D] and [hi, lo][DV][Nkp] per (sequence, head); row / column index includes k0
  _Float16* Kh = IMG ? p.kp + tile * I::KTI - (long)k0 * I::KP : p.kp + (long)blockIdx.y * 2 * p.Nkp * D;
  _Float16* Kl = IMG ? Kh + 32 * I::KP : Kh + (long)p.Nkp * D;
  _Float16* Vh = IMG ? p.vp + tile * I::VTI - k0 : p.vp + (long)blockIdx.y * 2 * DV * p.Nkp;
  _Float16* Vl = IMG ? Vh + DV * I::VP : Vh + (long)DV * p.Nkp;
  const long kpitch = IMG ? I::KP : D, vpitch = IMG ? I::VP : p.Nkp;
  for (int idx = tid; idx < 32 * (D / 4); idx += 256) {
    const int row = idx / (D / 4), c4 = idx - row * (D / 4);
    const bool ok = k0 + row < p.Nk;
    const int kr = ok ? (tix ? tix[k0 + row] : k0 + row) : 0;
    float4 kv = *reinterpret_cast<const float4*>(kb + (long)kr * p.k_sT + c4 * 4);
    kv = ok ? make_float4(kv.x * ATT_KSC, kv.y * ATT_KSC, kv.z * ATT_KSC, kv.w * ATT_KSC) : make_float4(0.f, 0.f, 0.f, 0.f);
    v4h hi, lo;
    split4(kv, hi, lo);
    *reinterpret_cast<v4h*>(&Kh[(long)(k0 + row) * kpitch + c4 * 4]) = hi;
    *reinterpret_cast<v4h*>(&Kl[(long)(k0 + row) * kpitch + c4 * 4]) = lo;
  }
  for (int idx = tid; idx < 16 * (DV / 4); idx += 256) {
    const int kp = idx & 15, c4 = idx >> 4;
    const bool ok0 = k0 + 2 * kp < p.Nk, ok1 = k0 + 2 * kp + 1 < p.Nk;
    const int r0 = ok0 ? (tix ? tix[k0 + 2 * kp] : k0 + 2 * kp) : 0;
    const int r1 = ok1 ? (tix ? tix[k0 + 2 * kp + 1] : k0 + 2 * kp + 1) : 0;
    float4 v0 = *reinterpret_cast<const float4*>(vb + (long)r0 * p.v_sT + c4 * 4);
    float4 v1 = *reinterpret_cast<const float4*>(vb + (long)r1 * p.v_sT + c4 * 4);
    v0 = ok0 ? make_float4(v0.x * ATT_VSC, v0.y * ATT_VSC, v0.z * ATT_VSC, v0.w * ATT_VSC) : make_float4(0.f, 0.f, 0.f, 0.f);
    v1 = ok1 ? make_float4(v1.x * ATT_VSC, v1.y * ATT_VSC, v1.z * ATT_VSC, v1.w * ATT_VSC) : make_float4(0.f, 0.f, 0.f, 0.f);
    v4h h0, l0, h1, l1;
    split4(v0, h0, l0);
    split4(v1, h1, l1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      typedef _Float16 v2h __attribute__((ext_vector_type(2)));
      v2h ph, pl;
      ph[0] = h0[j]; ph[1] = h1[j];
      pl[0] = l0[j]; pl[1] = l1[j];
      *reinterpret_cast<v2h*>(&Vh[(long)(c4 * 4 + j) * vpitch + k0 + 2 * kp]) = ph;
      *reinterpret_cast<v2h*>(&Vl[(long)(c4 * 4 + j) * vpitch + k0 + 2 * kp]) = pl;
    }
  }
}

template <int D, int DV, bool PL = false, bool PF = false>
__global__ __launch_bounds__(256, 2) void attn_f16x3_kernel(const AttnArgs p) {
  constexpr int KP = D + 8;     // K plane row pitch (halves)
  constexpr int VP = 36;        // Vt plane row pitch (halves): 32 keys + 4
  constexpr int NS = D / 16;    // k16 steps of S
  constexpr int NT = DV / 32;   // 32-wide output tiles along d
  // Power-of-two operand scales (exact; undone on the fp32 side).  The lo half of a split value is ~2^-12 of it, and
  // f16 loses precision below 2^-14 (subnormals; the matrix pipe may flush them): q*scale ~ 0.1 and p <= 1 would keep
  // only their hi halves.  Scaled, every operand of ordinary magnitude has a NORMAL lo half.
  constexpr float QSC = 64.f, KSC = ATT_KSC, PSC = ATT_PSC, VSC = ATT_VSC;
  // With prefetched planes (PL && PF) the K / V tiles are DOUBLE-buffered in LDS: tile t+1 is written (from the registers the
  // prefetch filled) behind tile t's MFMAs, so a key tile costs one barrier, not two, and no wave waits for staging.
  constexpr int TILE = 2 * 32 * KP + 2 * DV * VP;
  __shared__ __attribute__((aligned(16))) _Float16 smem[(PL && PF ? 2 : 1) * TILE];
  _Float16 *Kh = smem, *Kl = smem + 32 * KP, *Vh = smem + 2 * 32 * KP, *Vl = smem + 2 * 32 * KP + DV * VP;
  auto set_buf = [&](int bsel) {
    Kh = smem + bsel * TILE; Kl = Kh + 32 * KP; Vh = Kh + 2 * 32 * KP; Vl = Vh + DV * VP;
  };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y / p.Hh, h = blockIdx.y % p.Hh;
  const int q = blockIdx.x * 128 + wave * 32 + li;
  const bool qok = q < p.Nq;
  const int sample = p.tok_idx ? b / p.nwin : b;
  const int* tix = p.tok_idx ? p.tok_idx + (long)(b % p.nwin) * p.Nk : nullptr;
  const int qrow = qok ? (tix ? tix[q] : q) : 0;

  // Q^T fragments (B operand): lane (q, half) holds d = 16 s + 8 half + e
  v8h qh[NS], ql[NS];
  {
    const float* qp = p.q + (long)sample * p.q_sB + (long)h * p.q_sH + (long)qrow * p.q_sT + 8 * lh;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float4 a = *reinterpret_cast<const float4*>(qp + 16 * s);
      float4 c = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
      if (!qok) { a = make_float4(0.f, 0.f, 0.f, 0.f); c = a; }
      const float qs = QSC;   // the softmax scale is applied to S in fp32, so this scaling stays an exact power of two
      const float f[8] = {a.x * qs, a.y * qs, a.z * qs, a.w * qs, c.x * qs, c.y * qs, c.z * qs, c.w * qs};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        _Float16 hh, ll;
        split_f16(f[e], hh, ll);
        qh[s][e] = hh; ql[s][e] = ll;
      }
    }
  }

  v16f acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const float* kb = p.k + (long)sample * p.k_sB + (long)h * p.k_sH;
  const float* vb = p.v + (long)sample * p.v_sB + (long)h * p.v_sH;

  constexpr int PFK = PL ? (32 * (D / 8) + 255) / 256 : 1, PFV = PL ? (DV * 4 + 255) / 256 : 1;
  uint4 pf_kh[PFK], pf_kl[PFK], pf_vh[PFV], pf_vl[PFV];
  auto prefetch = [&](int k0) {
    if (!PL) return;
    const _Float16* gKh = p.kp + (long)blockIdx.y * 2 * p.Nkp * D + (long)k0 * D;
    const _Float16* gKl = gKh + (long)p.Nkp * D;
    const _Float16* gVh = p.vp + (long)blockIdx.y * 2 * DV * p.Nkp + k0;
    const _Float16* gVl = gVh + (long)DV * p.Nkp;
#pragma unroll
    for (int i = 0; i < PFK; ++i) {
      const int idx = tid + 256 * i;
      const int ii = idx < 32 * (D / 8) ? idx : 0;            // a K tile is one contiguous run of 32 * D halves per plane
      pf_kh[i] = *reinterpret_cast<const uint4*>(gKh + 8 * ii);
      pf_kl[i] = *reinterpret_cast<const uint4*>(gKl + 8 * ii);
    }
#pragma unroll
    for (int i = 0; i < PFV; ++i) {
      const int idx = tid + 256 * i;
      const int ii = idx < DV * 4 ? idx : 0;
      const int d = ii >> 2, c = ii & 3;
      pf_vh[i] = *reinterpret_cast<const uint4*>(gVh + (long)d * p.Nkp + 8 * c);
      pf_vl[i] = *reinterpret_cast<const uint4*>(gVl + (long)d * p.Nkp + 8 * c);
    }
  };
  // planes made by attn_kv_planes_kernel: staging is a copy from the prefetch registers -- 16 B per K chunk, 8 + 8 B per V^T
  // chunk (its LDS rows are 72 B apart), no conversion work
  auto stage_write = [&](_Float16* base) {
    _Float16 *bKh = base, *bKl = base + 32 * KP, *bVh = base + 2 * 32 * KP, *bVl = base + 2 * 32 * KP + DV * VP;
#pragma unroll
    for (int i = 0; i < PFK; ++i) {
      const int idx = tid + 256 * i;
      if (idx < 32 * (D / 8)) {
        const int row = idx / (D / 8), c = idx - row * (D / 8);
        *reinterpret_cast<uint4*>(&bKh[row * KP + 8 * c]) = pf_kh[i];
        *reinterpret_cast<uint4*>(&bKl[row * KP + 8 * c]) = pf_kl[i];
      }
    }
#pragma unroll
    for (int i = 0; i < PFV; ++i) {
      const int idx = tid + 256 * i;
      if (idx < DV * 4) {
        const int d = idx >> 2, c = idx & 3;
        *reinterpret_cast<uint2*>(&bVh[d * VP + 8 * c]) = make_uint2(pf_vh[i].x, pf_vh[i].y);
        *reinterpret_cast<uint2*>(&bVh[d * VP + 8 * c + 4]) = make_uint2(pf_vh[i].z, pf_vh[i].w);
        *reinterpret_cast<uint2*>(&bVl[d * VP + 8 * c]) = make_uint2(pf_vl[i].x, pf_vl[i].y);
        *reinterpret_cast<uint2*>(&bVl[d * VP + 8 * c + 4]) = make_uint2(pf_vl[i].z, pf_vl[i].w);
      }
    }
  };
  // key split: workgroup z of gridDim.z walks key tiles [kbeg, kend) (the host makes every slice non-empty)
  int kbeg = 0, kend = p.Nk;
  if (PL && PF && gridDim.z > 1) {
    const int ntile = (p.Nk + 31) >> 5, tps = (ntile + gridDim.z - 1) / gridDim.z;
    kbeg = blockIdx.z * tps * 32;
    kend = min(p.Nk, kbeg + tps * 32);
  }
  if (PL && PF) {
    prefetch(kbeg);
    stage_write(smem);
    if (kbeg + 32 < kend) prefetch(kbeg + 32);
    __syncthreads();
  }

  int tsel = 0;
  for (int k0 = kbeg; k0 < kend; k0 += 32, tsel ^= 1) {
    if (PL && PF) set_buf(tsel);
    if (!(PL && PF)) __syncthreads();  // previous tile fully consumed
    if (PL && !PF) prefetch(k0);
    if (PL && !PF) stage_write(smem);
    // K: thread -> (key row, 4 d): two 8-B plane writes
    for (int idx = tid; !PL && idx < 32 * (D / 4); idx += 256) {
      const int row = idx / (D / 4), c4 = idx - row * (D / 4);
      const bool ok = k0 + row < p.Nk;
      const int kr = ok ? (tix ? tix[k0 + row] : k0 + row) : 0;
      float4 kv = *reinterpret_cast<const float4*>(kb + (long)kr * p.k_sT + c4 * 4);
      kv = ok ? make_float4(kv.x * KSC, kv.y * KSC, kv.z * KSC, kv.w * KSC) : make_float4(0.f, 0.f, 0.f, 0.f);
      v4h hi, lo;
      split4(kv, hi, lo);
      *reinterpret_cast<v4h*>(&Kh[row * KP + c4 * 4]) = hi;
      *reinterpret_cast<v4h*>(&Kl[row * KP + c4 * 4]) = lo;
    }
    // V, transposed: thread -> (key pair fastest, 4 d): 4-B writes Vt[d][2kp .. 2kp+1]
    for (int idx = tid; !PL && idx < 16 * (DV / 4); idx += 256) {
      const int kp = idx & 15, c4 = idx >> 4;
      const bool ok0 = k0 + 2 * kp < p.Nk, ok1 = k0 + 2 * kp + 1 < p.Nk;
      const int r0 = ok0 ? (tix ? tix[k0 + 2 * kp] : k0 + 2 * kp) : 0;
      const int r1 = ok1 ? (tix ? tix[k0 + 2 * kp + 1] : k0 + 2 * kp + 1) : 0;
      float4 v0 = *reinterpret_cast<const float4*>(vb + (long)r0 * p.v_sT + c4 * 4);
      float4 v1 = *reinterpret_cast<const float4*>(vb + (long)r1 * p.v_sT + c4 * 4);
      v0 = ok0 ? make_float4(v0.x * VSC, v0.y * VSC, v0.z * VSC, v0.w * VSC) : make_float4(0.f, 0.f, 0.f, 0.f);
      v1 = ok1 ? make_float4(v1.x * VSC, v1.y * VSC, v1.z * VSC, v1.w * VSC) : make_float4(0.f, 0.f, 0.f, 0.f);
      v4h h0, l0, h1, l1;
      split4(v0, h0, l0);
      split4(v1, h1, l1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        typedef _Float16 v2h __attribute__((ext_vector_type(2)));
        v2h ph, pl;
        ph[0] = h0[j]; ph[1] = h1[j];
        pl[0] = l0[j]; pl[1] = l1[j];
        *reinterpret_cast<v2h*>(&Vh[(c4 * 4 + j) * VP + 2 * kp]) = ph;
        *reinterpret_cast<v2h*>(&Vl[(c4 * 4 + j) * VP + 2 * kp]) = pl;
      }
    }
    if (!(PL && PF)) __syncthreads();

    // S^T = K . Q^T
    v16f s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const v8h kh = *reinterpret_cast<const v8h*>(&Kh[li * KP + 16 * st + 8 * lh]);
      const v8h kl = *reinterpret_cast<const v8h*>(&Kl[li * KP + 16 * st + 8 * lh]);
      if (!kSingleProduct) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[st], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[st], s, 0, 0, 0);
      }
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[st], s, 0, 0, 0);
    }
#pragma unroll
    // logits in LOG2 units: log2(e) rides on the scale, the exponentials below are bare v_exp_f32
    for (int r = 0; r < 16; ++r) s[r] *= p.scale * (1.4426950408889634f / (QSC * KSC));

    // online softmax over the 32 keys of this tile (16 in my registers, 16 in lane^32's)
    float mt = -INFINITY;
    if (p.biasT || p.maskT) {
      const float* bt = p.biasT ? p.biasT + (long)h * p.Nk * p.Nq : nullptr;
      const float* mk = p.maskT ? p.maskT + (long)(b % p.nmask) * p.Nk * p.Nq : nullptr;
      float add[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const long o = (qok && key < p.Nk) ? (long)key * p.Nq + q : 0;
        add[r] = (bt ? bt[o] : 0.f) + (mk ? mk[o] : 0.f);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fmaf(add[r], 1.4426950408889634f, s[r]);
    }
    if (k0 + 32 > p.Nk) {      // only the last tile has keys past the end (wave-uniform branch)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= p.Nk) s[r] = -INFINITY;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float ps = 0.f;
    v8h ph[2], pl[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(s[r] - m_new);
      ps += e;
      _Float16 hi, lo;
      split_f16(e * PSC, hi, lo);
      ph[r >> 3][r & 7] = hi;
      pl[r >> 3][r & 7] = lo;
    }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {      // no row of this wave raised its maximum: nothing to rescale
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;
    }

    // O^T += V^T . P^T
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int d = t * 32 + li;
        const v4h a0 = *reinterpret_cast<const v4h*>(&Vh[d * VP + 16 * s2 + 4 * lh]);
        const v4h a1 = *reinterpret_cast<const v4h*>(&Vh[d * VP + 16 * s2 + 8 + 4 * lh]);
        const v4h c0 = *reinterpret_cast<const v4h*>(&Vl[d * VP + 16 * s2 + 4 * lh]);
        const v4h c1 = *reinterpret_cast<const v4h*>(&Vl[d * VP + 16 * s2 + 8 + 4 * lh]);
        const v8h vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const v8h vl = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        // -DMSPI_ATT_PV_DROP=1 / 2 / 3: measurement builds without the V_hi.P_lo / V_lo.P_hi / both cross products of
        // O^T += V^T.P^T (which of them the 1e-3 bar on the map tolerates)
        if (!kSingleProduct && !(MSPI_ATT_PV_DROP & 2)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s2], acc[t], 0, 0, 0);
        if (!kSingleProduct && !(MSPI_ATT_PV_DROP & 1)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s2], acc[t], 0, 0, 0);
      }
    if (PL && PF) {
      if (k0 + 32 < kend) {
        stage_write(smem + (tsel ^ 1) * TILE);                 // tile t+1 (fetched during this tile) into the other buffer
        if (k0 + 64 < kend) prefetch(k0 + 64);
      }
      __syncthreads();                                         // tile t consumed by every wave, tile t+1 visible
    }
  }

  if (PL && PF && gridDim.z > 1) {
    if (qok) {
      const long pr = ((long)blockIdx.z * gridDim.y + blockIdx.y) * p.Nq + q;
      if (lh == 0) *reinterpret_cast<float2*>(p.part_ml + pr * 2) = make_float2(m_run, l_run);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(p.part_o + pr * DV + t * 32 + 8 * g + 4 * lh) =
              make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
    }
    return;
  }
  if (qok) {
    const float inv = 1.f / (l_run * (PSC * VSC));
    const long oo = (long)sample * p.o_sB + (long)h * p.o_sH + (long)qrow * p.o_sT;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 o4 = make_float4(acc[t][4 * g] * inv, acc[t][4 * g + 1] * inv, acc[t][4 * g + 2] * inv,
                                acc[t][4 * g + 3] * inv);
        const int dd = t * 32 + 8 * g + 4 * lh;
        if (p.res) {
          const float4 rr = *reinterpret_cast<const float4*>(p.res + oo + dd);
          o4.x += rr.x; o4.y += rr.y; o4.z += rr.z; o4.w += rr.w;
        }
        *reinterpret_cast<float4*>(p.o + oo + dd) = o4;
      }
  }
}

// ------------------------------------------------------------------ the same kernel as a software pipeline over the key tiles
// attn_f16x3_kernel alternates, per wave, an MFMA phase (S), a VALU phase (softmax) and an MFMA phase (PV).  On this chip a
// wave's own VALU instructions issue in the shadow of its MFMAs for free, while an MFMA-phase wave and a VALU-phase wave that
// share a SIMD slow each other to less than the sum (tools/coissue_probe.hip, profiles/r03_coissue_probe.txt) -- so the
// second workgroup per CU hides almost nothing (a CU retires one 128 x 32 step per ~3.3 us however many share it).
// Here iteration t holds, in one hand-interleaved instruction stream per wave,
//     S of tile t+1 (MFMA)  ||  softmax of tile t (VALU)      then      PV of tile t (MFMA)  ||  staging of K(t+2), V(t+1)
// Each MFMA triple (3 split products on one operand pair) is followed by a slice of the VALU / staging work and closed by a
// sched_barrier so that the order survives the compiler; operand fragments are read from LDS one triple ahead.  K tiles run
// one tile ahead of V tiles in the LDS double buffers.  Planes + prefetch path only (no bias, mask or token index); the
// arithmetic per query row is that of attn_f16x3_kernel (same products, same order), so results are bit-identical.
template <int D, int DV>
__global__ __launch_bounds__(256, 2) void attn_pipe_kernel(const AttnArgs p) {
  constexpr int KP = D + 8, VP = 36, NS = D / 16, NT = DV / 32;
  constexpr float QSC = 64.f, KSC = ATT_KSC, PSC = ATT_PSC, VSC = ATT_VSC;
  typedef AttnImg<D, DV> I;
  static_assert(I::KP == KP && I::VP == VP, "LDS layout = plane image layout");
  constexpr int KT = I::KTI, VT = I::VTI;                 // halves per staged K / V tile (hi + lo planes) = the plane images
  __shared__ __attribute__((aligned(16))) _Float16 smem[2 * KT + 2 * VT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y / p.Hh, h = blockIdx.y % p.Hh;
  const int q = blockIdx.x * 128 + wave * 32 + li;
  const bool qok = q < p.Nq;

  v8h qh[NS], ql[NS];
  {
    const float* qp = p.q + (long)b * p.q_sB + (long)h * p.q_sH + (long)(qok ? q : 0) * p.q_sT + 8 * lh;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float4 a = *reinterpret_cast<const float4*>(qp + 16 * s);
      float4 c = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
      if (!qok) { a = make_float4(0.f, 0.f, 0.f, 0.f); c = a; }
      const float f[8] = {a.x * QSC, a.y * QSC, a.z * QSC, a.w * QSC, c.x * QSC, c.y * QSC, c.z * QSC, c.w * QSC};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        _Float16 hh, ll;
        split_f16(f[e], hh, ll);
        qh[s][e] = hh; ql[s][e] = ll;
      }
    }
  }
  v16f acc[NT], sc, scn;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f, m_new = 0.f, alpha = 1.f, psum = 0.f;
  v8h ph[2], pl[2];

  // a tile image is staged as whole 1-KB LDS-DMA pieces (global_load_lds, 16 B per lane), dealt round-robin to the 4 waves:
  // no registers, no LDS writes, and the data lands while the wave computes
  const _Float16* const gK = p.kp + (long)blockIdx.y * (p.Nkp >> 5) * KT;
  const _Float16* const gV = p.vp + (long)blockIdx.y * (p.Nkp >> 5) * VT;
  auto dma_k = [&](int tile, _Float16* dst) {
    const _Float16* src = gK + (long)tile * KT;
#pragma unroll
    for (int j = 0; j < (KT / 512 + 3) / 4; ++j) {
      const int pc = wave + 4 * j;
      if (pc < KT / 512)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + pc * 512 + lane * 8), (lds_void*)(dst + pc * 512), 16, 0, 0);
    }
  };
  auto dma_v = [&](int tile, _Float16* dst) {
    const _Float16* src = gV + (long)tile * VT;
#pragma unroll
    for (int j = 0; j < (VT / 512 + 3) / 4; ++j) {
      const int pc = wave + 4 * j;
      if (pc < VT / 512)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + pc * 512 + lane * 8), (lds_void*)(dst + pc * 512), 16, 0, 0);
    }
  };

  int kbeg = 0, kend = p.Nk;
  if (gridDim.z > 1) {
    const int ntile = (p.Nk + 31) >> 5, tps = (ntile + gridDim.z - 1) / gridDim.z;
    kbeg = blockIdx.z * tps * 32;
    kend = min(p.Nk, kbeg + tps * 32);
  }
  const int nt = (kend - kbeg + 31) >> 5;
  _Float16* const Kb = smem;
  _Float16* const Vb = smem + 2 * KT;
  const int t0 = kbeg >> 5;
  // prologue: K(0), V(0), K(1) staged
  dma_k(t0, Kb);
  dma_v(t0, Vb);
  if (nt > 1) dma_k(t0 + 1, Kb + KT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float sscale = p.scale * (1.4426950408889634f / (QSC * KSC));
  // S^T = K . Q^T from a staged K tile (NS triples); behind triple st runs slice(st)
  auto region_s = [&](v16f& s, const _Float16* K, auto&& slice) {
    const _Float16* Kl = K + 32 * KP;
    // fragments are read TWO triples ahead: a ds_read_b128 issued one triple (96 cycles) ahead is not back in time
    auto kf = [&](int st, const _Float16* P) { return *reinterpret_cast<const v8h*>(&P[li * KP + 16 * st + 8 * lh]); };
    v8h kh = kf(0, K), kl = kf(0, Kl), kh1 = kh, kl1 = kl;
    if (NS > 1) { kh1 = kf(1, K); kl1 = kf(1, Kl); }
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      v8h nh = kh1, nl = kl1;
      if (st + 2 < NS) { nh = kf(st + 2, K); nl = kf(st + 2, Kl); }
      __builtin_amdgcn_sched_barrier(0);
      const v16f zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (!kSingleProduct) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[st], st == 0 ? zero : s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[st], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[st], s, 0, 0, 0);
      } else {
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[st], st == 0 ? zero : s, 0, 0, 0);
      }
      slice(st);
      kh = kh1; kl = kl1; kh1 = nh; kl1 = nl;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // slice i of the online softmax over this tile's 32 keys (16 in my registers, 16 in lane^32's): 0 scale (+ mask of the keys
  // past the end), 1 running maximum and rescale factor, 2..17 one probability each (exp2, sum, split), 18 the sum
  // Slices 0 .. NP1-1 (the probabilities of the first 16-key half included) run behind the S triples; the second half's
  // run behind the first NT PV triples, which only need the first half's probabilities -- the S region is VALU-bound
  // (~1050 cycles of softmax against 768 of MFMA at D = 128), the PV region has no other VALU work.
  constexpr int NPC = 19, NP1 = 10;
  auto sm_piece = [&](int i, int k0) {
    if (i == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] *= sscale;
      if (k0 + 32 > p.Nk) {      // only the last tile has keys past the end (wave-uniform branch)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= p.Nk) sc[r] = -INFINITY;
      }
    } else if (i == 1) {
      float mt = sc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sc[r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      m_new = fmaxf(m_run, mt);
      alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      psum = 0.f;
    } else if (i < 18) {
      const int r = i - 2;
      const float e = __builtin_amdgcn_exp2f(sc[r] - m_new);
      psum += e;
      _Float16 hi, lo;
      split_f16(e * PSC, hi, lo);
      ph[r >> 3][r & 7] = hi;
      pl[r >> 3][r & 7] = lo;
    } else {
      const float ps = psum + __shfl_xor(psum, 32, 64);
      l_run = l_run * alpha + ps;
    }
  };

  region_s(sc, Kb, [](int) {});
  __syncthreads();            // every wave is done with K(0): iteration 0 overwrites its buffer with K(2)
  for (int t = 0; t < nt; ++t) {
    const int k0 = kbeg + 32 * t;
    // ---- S of tile t+1 with the softmax of tile t behind its triples
    if (t + 1 < nt) {
      region_s(scn, Kb + ((t + 1) & 1) * KT, [&](int st) {
#pragma unroll
        for (int i = st * NP1 / NS; i < (st + 1) * NP1 / NS; ++i) sm_piece(i, k0);
      });
    } else {
#pragma unroll
      for (int i = 0; i < NP1; ++i) sm_piece(i, k0);
    }
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {      // no row of this wave raised its maximum: nothing to rescale
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] *= alpha;
    }
    // K(t+2) -> the buffer of K(t) (consumed by S(t) in iteration t-1), V(t+1) -> the buffer of V(t-1): issued here, waited
    // for at the end of this iteration, first read in iteration t+1
    // ---- PV of tile t
    {
      const _Float16* Vh = Vb + (t & 1) * VT;
      const _Float16* Vl = Vh + DV * VP;
      auto vfrag = [&](int tr, const _Float16* V) {
        const int s2 = tr / NT, u = tr % NT, d = u * 32 + li;
        const v4h a0 = *reinterpret_cast<const v4h*>(&V[d * VP + 16 * s2 + 4 * lh]);
        const v4h a1 = *reinterpret_cast<const v4h*>(&V[d * VP + 16 * s2 + 8 + 4 * lh]);
        const v8h v = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        return v;
      };
      v8h vh = vfrag(0, Vh), vl = vfrag(0, Vl), vh1 = vh, vl1 = vl;
      if (2 * NT > 1) { vh1 = vfrag(1, Vh); vl1 = vfrag(1, Vl); }
#pragma unroll
      for (int tr = 0; tr < 2 * NT; ++tr) {
        v8h nh = vh1, nl = vl1;
        if (tr + 2 < 2 * NT) { nh = vfrag(tr + 2, Vh); nl = vfrag(tr + 2, Vl); }
        __builtin_amdgcn_sched_barrier(0);
        const int s2 = tr / NT, u = tr % NT;
        if (!kSingleProduct) {
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s2], acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s2], acc[u], 0, 0, 0);
        }
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s2], acc[u], 0, 0, 0);
        if (tr < NT) {
#pragma unroll
          for (int i = NP1 + tr * (NPC - NP1) / NT; i < NP1 + (tr + 1) * (NPC - NP1) / NT; ++i) sm_piece(i, k0);
        }
        if (tr == 0 && t + 2 < nt) dma_k(t0 + t + 2, Kb + (t & 1) * KT);
        if (tr == (2 * NT > 1 ? 1 : 0) && t + 1 < nt) dma_v(t0 + t + 1, Vb + ((t + 1) & 1) * VT);
        vh = vh1; vl = vl1; vh1 = nh; vl1 = nl;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of K(t+2), V(t+1) have landed
    __syncthreads();          // tile t consumed by every wave; K(t+2), V(t+1) visible
    sc = scn;
  }

  if (gridDim.z > 1) {
    if (qok) {
      const long pr = ((long)blockIdx.z * gridDim.y + blockIdx.y) * p.Nq + q;
      if (lh == 0) *reinterpret_cast<float2*>(p.part_ml + pr * 2) = make_float2(m_run, l_run);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(p.part_o + pr * DV + t * 32 + 8 * g + 4 * lh) =
              make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
    }
    return;
  }
  if (qok) {
    const float inv = 1.f / (l_run * (PSC * VSC));
    const long oo = (long)b * p.o_sB + (long)h * p.o_sH + (long)q * p.o_sT;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 o4 = make_float4(acc[t][4 * g] * inv, acc[t][4 * g + 1] * inv, acc[t][4 * g + 2] * inv,
                                acc[t][4 * g + 3] * inv);
        const int dd = t * 32 + 8 * g + 4 * lh;
        if (p.res) {
          const float4 rr = *reinterpret_cast<const float4*>(p.res + oo + dd);
          o4.x += rr.x; o4.y += rr.y; o4.z += rr.z; o4.w += rr.w;
        }
        *reinterpret_cast<float4*>(p.o + oo + dd) = o4;
      }
  }
}

// Key split, second pass: the slices' partial results are merged in fixed order z = 0, 1, .. (deterministic):
//   M = max_z m_z,  w_z = 2^(m_z - M),  O = sum_z w_z O_z / (PSC VSC sum_z w_z l_z)  (+ res)
template <int DV>
__global__ __launch_bounds__(256) void attn_merge_kernel(const AttnArgs p, int nz) {
  const long BH = (long)p.B * p.Hh;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= BH * p.Nq * (DV / 4)) return;
  const int c4 = (int)(idx % (DV / 4));
  const long row = idx / (DV / 4);          // bh * Nq + q
  const int q = (int)(row % p.Nq);
  const long bh = row / p.Nq;
  float M = -INFINITY;
  for (int z = 0; z < nz; ++z) M = fmaxf(M, p.part_ml[((long)z * BH * p.Nq + row) * 2]);
  float L = 0.f;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < nz; ++z) {
    const float2 ml = *reinterpret_cast<const float2*>(p.part_ml + ((long)z * BH * p.Nq + row) * 2);
    const float w = __builtin_amdgcn_exp2f(ml.x - M);
    L = fmaf(ml.y, w, L);
    const float4 a = *reinterpret_cast<const float4*>(p.part_o + ((long)z * BH * p.Nq + row) * DV + c4 * 4);
    o.x = fmaf(a.x, w, o.x); o.y = fmaf(a.y, w, o.y); o.z = fmaf(a.z, w, o.z); o.w = fmaf(a.w, w, o.w);
  }
  const float inv = 1.f / (L * (ATT_PSC * ATT_VSC));
  o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
  const long oo = (bh / p.Hh) * p.o_sB + (bh % p.Hh) * p.o_sH + (long)q * p.o_sT + c4 * 4;
  if (p.res) {
    const float4 rr = *reinterpret_cast<const float4*>(p.res + oo);
    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
  }
  *reinterpret_cast<float4*>(p.o + oo) = o;
}

// ------------------------------------------------------------------ MViTv2 decomposed relative positions
// attn = (q*scale) k^T + q.Rh[hq,hk] + q.Rw[wq,wk] + q.Rt[tq,tk]   (backbones/MViT.py:905-997,1261-1290).
// The three rank-structured terms are folded into the contraction itself: Q gets J = kH+kW+kT extra columns
// holding q.R*(own position, j) and K gets the matching one-hot columns, so one fused attention pass over
// D + J columns reproduces the sum exactly (a 1.0 x value product is exact) -- no bias matrix exists anywhere.
struct AugArgs {
  const float* q;   // pooled + normed q rows [B*Nq][ldq], head h at column h*Dh
  const float* k;   // pooled + normed k rows [B*Nk][ldk]
  const float* Rh;  // [qH][kH][Dh] gathered tables
  const float* Rw;  // [qW][kW][Dh]
  const float* Rt;  // [qT][kT][Dh]
  float* qa;        // [B][heads][Nq][DA]
  float* ka;        // [B][heads][Nk][DA]
  int B, heads, Dh, DA;
  int qT, qH, qW, kT, kH, kW;
  long ldq, ldk;
  float scale;
};

// (1) copy part, 16-B vectors: qa[.., 0:Dh] = scale*q, ka[.., 0:Dh] = k, one-hot columns of ka, zero tail of both
__global__ __launch_bounds__(256) void mvit_aug_copy_kernel(const AugArgs p) {
  const int Nq = p.qT * p.qH * p.qW, Nk = p.kT * p.kH * p.kW;
  const int J = p.kH + p.kW + p.kT, DV4 = p.DA >> 2, DH4 = p.Dh >> 2;
  const long qrows = (long)p.B * p.heads * Nq, krows = (long)p.B * p.heads * Nk;
  const long total = (qrows + krows) * DV4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c4 = (int)(idx % DV4);
    long row = idx / DV4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < qrows) {
      if (c4 >= DH4 && c4 * 4 < p.Dh + J) continue;   // relative-position columns: written by mvit_aug_rel_kernel
      const int tok = (int)(row % Nq);
      const long bh = row / Nq;
      if (c4 < DH4) {
        v = *reinterpret_cast<const float4*>(p.q + ((bh / p.heads) * Nq + tok) * p.ldq + (bh % p.heads) * p.Dh + c4 * 4);
        v.x *= p.scale; v.y *= p.scale; v.z *= p.scale; v.w *= p.scale;
      }
      *reinterpret_cast<float4*>(p.qa + row * p.DA + c4 * 4) = v;
    } else {
      row -= qrows;
      const int tok = (int)(row % Nk);
      const long bh = row / Nk;
      if (c4 < DH4) {
        v = *reinterpret_cast<const float4*>(p.k + ((bh / p.heads) * Nk + tok) * p.ldk + (bh % p.heads) * p.Dh + c4 * 4);
      } else {
        const int wk = tok % p.kW, hk = (tok / p.kW) % p.kH, tk = tok / (p.kW * p.kH);
        const int j0 = c4 * 4 - p.Dh, a = hk, b = p.kH + wk, c = p.kH + p.kW + tk;
        v.x = (j0 == a || j0 == b || j0 == c) ? 1.f : 0.f;
        v.y = (j0 + 1 == a || j0 + 1 == b || j0 + 1 == c) ? 1.f : 0.f;
        v.z = (j0 + 2 == a || j0 + 2 == b || j0 + 2 == c) ? 1.f : 0.f;
        v.w = (j0 + 3 == a || j0 + 3 == b || j0 + 3 == c) ? 1.f : 0.f;
      }
      *reinterpret_cast<float4*>(p.ka + row * p.DA + c4 * 4) = v;
    }
  }
}

// (2) the J dot products per query row: a workgroup = 8 rows x 32 lanes; lane j < J computes q_row . table_j
// (q loads are wave broadcasts, the tables are a few KB and stay in L1).  Also zero-fills up to the next 4-column
// boundary so that the copy kernel's vectors and these scalars tile the row exactly.
template <int DH>
__global__ __launch_bounds__(256) void mvit_aug_rel_kernel(const AugArgs p) {
  const int Nq = p.qT * p.qH * p.qW;
  const int J = p.kH + p.kW + p.kT, J4 = (p.Dh + J + 3) / 4 * 4 - p.Dh;
  const long qrows = (long)p.B * p.heads * Nq;
  const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int j = threadIdx.x & 31;
  if (row >= qrows) return;
  const int tok = (int)(row % Nq);
  const long bh = row / Nq;
  const float* qr = p.q + ((bh / p.heads) * Nq + tok) * p.ldq + (bh % p.heads) * p.Dh;
  const int wq = tok % p.qW, hq = (tok / p.qW) % p.qH, tq = tok / (p.qW * p.qH);
  for (int jj = j; jj < J4; jj += 32) {
    float v = 0.f;
    if (jj < J) {
      const float* tab = jj < p.kH ? p.Rh + ((long)hq * p.kH + jj) * p.Dh
                       : jj < p.kH + p.kW ? p.Rw + ((long)wq * p.kW + (jj - p.kH)) * p.Dh
                                          : p.Rt + ((long)tq * p.kT + (jj - p.kH - p.kW)) * p.Dh;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {   // DH is a compile-time constant: all 2*DH/4 loads are in flight together
        const float4 a = *reinterpret_cast<const float4*>(qr + c);
        const float4 t = *reinterpret_cast<const float4*>(tab + c);
        a0 = fmaf(a.x, t.x, a0); a1 = fmaf(a.y, t.y, a1); a2 = fmaf(a.z, t.z, a2); a3 = fmaf(a.w, t.w, a3);
      }
      v = (a0 + a1) + (a2 + a3);
    }
    p.qa[row * p.DA + p.Dh + jj] = v;
  }
}

// (2') the same J columns gathered from P = Q . [all distinct relative-position rows]^T, a dense thin GEMM the caller ran
// before (mspi_mvit_qk_augment_p): q . R*[own position, j] = P[row][index of the distance (own position, j)] -- the
// 96-long dot products move onto the matrix pipe and this kernel only moves 4 bytes per (row, j).
struct GatherArgs {
  const float* P;        // [(b*Nq + tok)*heads + head][ldp]
  const int* idx_h;      // [qH][kH] column of P
  const int* idx_w;      // [qW][kW]
  const int* idx_t;      // [qT][kT]
  float* qa;
  long ldp;
  int B, heads, Dh, DA, qT, qH, qW, kT, kH, kW;
};

__global__ __launch_bounds__(256) void mvit_aug_gather_kernel(const GatherArgs p) {
  const int Nq = p.qT * p.qH * p.qW;
  const int J = p.kH + p.kW + p.kT, J4 = (p.Dh + J + 3) / 4 * 4 - p.Dh;
  const long qrows = (long)p.B * p.heads * Nq;
  const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int j = threadIdx.x & 31;
  if (row >= qrows) return;
  const int tok = (int)(row % Nq);
  const long bh = row / Nq;
  const float* pr = p.P + (((bh / p.heads) * Nq + tok) * p.heads + bh % p.heads) * p.ldp;
  const int wq = tok % p.qW, hq = (tok / p.qW) % p.qH, tq = tok / (p.qW * p.qH);
  for (int jj = j; jj < J4; jj += 32) {
    float v = 0.f;
    if (jj < J) {
      const int col = jj < p.kH ? p.idx_h[hq * p.kH + jj]
                    : jj < p.kH + p.kW ? p.idx_w[wq * p.kW + (jj - p.kH)] : p.idx_t[tq * p.kT + (jj - p.kH - p.kW)];
      v = pr[col];
    }
    p.qa[row * p.DA + p.Dh + jj] = v;
  }
}

}  // namespace mspi

using namespace mspi;

extern "C" int mspi_mvit_qk_augment_p(const MspiMvitAugDesc* d, const float* q, const float* k, const float* P, int64_t ldp,
                                      const int32_t* idx_h, const int32_t* idx_w, const int32_t* idx_t, float* qa, float* ka,
                                      mspi_stream_t stream) {
  MSPI_REQUIRE(d && q && k && P && idx_h && idx_w && idx_t && qa && ka, "mspi_mvit_qk_augment_p: null argument");
  MSPI_REQUIRE(d->B > 0 && d->heads > 0 && d->Dh > 0 && (d->Dh & 3) == 0 && (d->DA & 3) == 0 &&
                   d->DA >= d->Dh + d->kH + d->kW + d->kT, "mspi_mvit_qk_augment_p: DA=%d too small", d->DA);
  MSPI_REQUIRE((d->ldq & 3) == 0 && (d->ldk & 3) == 0 && aligned16(q) && aligned16(k), "mspi_mvit_qk_augment_p: 16-B alignment");
  AugArgs a;
  a.q = q; a.k = k; a.Rh = nullptr; a.Rw = nullptr; a.Rt = nullptr; a.qa = qa; a.ka = ka;
  a.B = d->B; a.heads = d->heads; a.Dh = d->Dh; a.DA = d->DA;
  a.qT = d->qT; a.qH = d->qH; a.qW = d->qW; a.kT = d->kT; a.kH = d->kH; a.kW = d->kW;
  a.ldq = d->ldq; a.ldk = d->ldk; a.scale = d->scale;
  const long qrows = (long)d->B * d->heads * d->qT * d->qH * d->qW;
  const long total = (qrows + (long)d->B * d->heads * d->kT * d->kH * d->kW) * (d->DA / 4);
  long g = (total + 255) / 256;
  if (g > 256L * 32) g = 256L * 32;
  MSPI_REQUIRE((qrows + 7) / 8 < (1L << 31), "mspi_mvit_qk_augment_p: too many rows");
  hipLaunchKernelGGL(mvit_aug_copy_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
  GatherArgs ga;
  ga.P = P; ga.idx_h = idx_h; ga.idx_w = idx_w; ga.idx_t = idx_t; ga.qa = qa; ga.ldp = ldp;
  ga.B = d->B; ga.heads = d->heads; ga.Dh = d->Dh; ga.DA = d->DA;
  ga.qT = d->qT; ga.qH = d->qH; ga.qW = d->qW; ga.kT = d->kT; ga.kH = d->kH; ga.kW = d->kW;
  hipLaunchKernelGGL(mvit_aug_gather_kernel, dim3((unsigned)((qrows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, ga);
  return check_launch("mspi_mvit_qk_augment_p");
}

extern "C" int mspi_mvit_qk_augment(const MspiMvitAugDesc* d, const float* q, const float* k, const float* Rh,
                                    const float* Rw, const float* Rt, float* qa, float* ka, mspi_stream_t stream) {
  MSPI_REQUIRE(d && q && k && Rh && Rw && Rt && qa && ka, "mspi_mvit_qk_augment: null argument");
  MSPI_REQUIRE(d->B > 0 && d->heads > 0 && d->Dh > 0 && (d->Dh & 3) == 0 && d->DA >= d->Dh + d->kH + d->kW + d->kT,
               "mspi_mvit_qk_augment: DA=%d too small for Dh=%d + %d relative-position columns", d->DA, d->Dh,
               d->kH + d->kW + d->kT);
  MSPI_REQUIRE((d->ldq & 3) == 0 && (d->ldk & 3) == 0 && aligned16(q) && aligned16(k) && aligned16(Rh) && aligned16(Rw) &&
                   aligned16(Rt), "mspi_mvit_qk_augment: 16-B alignment");
  AugArgs a;
  a.q = q; a.k = k; a.Rh = Rh; a.Rw = Rw; a.Rt = Rt; a.qa = qa; a.ka = ka;
  a.B = d->B; a.heads = d->heads; a.Dh = d->Dh; a.DA = d->DA;
  a.qT = d->qT; a.qH = d->qH; a.qW = d->qW; a.kT = d->kT; a.kH = d->kH; a.kW = d->kW;
  a.ldq = d->ldq; a.ldk = d->ldk; a.scale = d->scale;
  MSPI_REQUIRE((d->DA & 3) == 0, "mspi_mvit_qk_augment: DA must be a multiple of 4");
  const long qrows = (long)d->B * d->heads * d->qT * d->qH * d->qW;
  const long total = (qrows + (long)d->B * d->heads * d->kT * d->kH * d->kW) * (d->DA / 4);
  long g = (total + 255) / 256;
  if (g > 256L * 32) g = 256L * 32;
  MSPI_REQUIRE((qrows + 7) / 8 < (1L << 31), "mspi_mvit_qk_augment: too many rows");
  hipLaunchKernelGGL(mvit_aug_copy_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
  MSPI_REQUIRE(d->Dh == 96, "mspi_mvit_qk_augment: head_dim %d not instantiated (MViTv2 uses 96)", d->Dh);
  hipLaunchKernelGGL(mvit_aug_rel_kernel<96>, dim3((unsigned)((qrows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("mspi_mvit_qk_augment");
}

static long attn_nkp(const MspiAttnDesc* d) { return ((long)d->Nk + 31) / 32 * 32; }

// Key split for few-query shapes (MViTv2's last stage: 8 heads x 392 queries = 256 workgroups of up to 49 key tiles each, one
// per CU at half occupancy): slices of the key tiles go to gridDim.z workgroups and a merge pass combines them.  Chosen so
// that the grid reaches two workgroups per CU and no slice is shorter than 12 tiles (measured, batch 8: Nk = 1568 195.8 ->
// 173.0 us, but Nk = 392 in two slices of 7 tiles 66.7 -> 71.6 us: the merge pass costs more than the second workgroup per
// CU gains -- a CU retires one 128 x 32 tile step per ~3.3 us however many workgroups share it); 1 = no split.
// MSPI_ATTN_KSPLIT=0: off.
static int attn_ksplit(const MspiAttnDesc* d) {
  static const char* env = getenv("MSPI_ATTN_KSPLIT");
  if (env && env[0] == '0') return 1;
  const long wgs = (((long)d->Nq + 127) / 128) * d->B * d->Hh;
  const int ntile = (int)(attn_nkp(d) / 32);
  int split = (int)(512 / (wgs > 0 ? wgs : 1));
  if (split > ntile / 12) split = ntile / 12;
  if (split > 8) split = 8;
  while (split > 1 && (split - 1) * ((ntile + split - 1) / split) >= ntile) --split;    // every slice non-empty
  return split < 2 ? 1 : split;
}

// plane bytes: the larger of the plain layout and the per-tile image layout (attn_pipe_kernel; AttnImg<D, DV>)
static size_t attn_planes_bytes(const MspiAttnDesc* d) {
  const size_t plain = (size_t)d->B * d->Hh * 2 * attn_nkp(d) * (size_t)(d->D + d->Dv) * sizeof(_Float16);
  const size_t kti = 2 * 32 * (size_t)(d->D + 8), vti = (2 * (size_t)d->Dv * 36 * 2 + 1023) / 1024 * 512;
  const size_t img = (size_t)d->B * d->Hh * (attn_nkp(d) / 32) * (kti + vti) * sizeof(_Float16);
  const size_t b = plain > img ? plain : img;
  return (b + 15) & ~(size_t)15;
}

extern "C" size_t mspi_attn_ws_bytes(const MspiAttnDesc* d) {
  if (!d || d->prec != MSPI_PREC_F16X3 || d->B <= 0 || d->Hh <= 0 || d->Nk <= 0) return 0;
  const int split = attn_ksplit(d);
  const size_t part = split > 1 ? (size_t)split * d->B * d->Hh * d->Nq * (size_t)(d->Dv + 2) * sizeof(float) : 0;
  return attn_planes_bytes(d) + part;
}

static int attn_fwd_impl(const MspiAttnDesc* d, const float* q, const float* k, const float* v, const float* res,
                         const float* biasT, const float* maskT, const int32_t* tok_idx, float* o, void* ws,
                         mspi_stream_t stream);

extern "C" int mspi_attn_fwd(const MspiAttnDesc* d, const float* q, const float* k, const float* v, const float* res,
                             const float* biasT, const float* maskT, const int32_t* tok_idx, float* o,
                             mspi_stream_t stream) {
  return attn_fwd_impl(d, q, k, v, res, biasT, maskT, tok_idx, o, nullptr, stream);
}

extern "C" int mspi_attn_fwd_ws(const MspiAttnDesc* d, const float* q, const float* k, const float* v, const float* res,
                                const float* biasT, const float* maskT, const int32_t* tok_idx, float* o, void* workspace,
                                mspi_stream_t stream) {
  MSPI_REQUIRE(workspace && aligned16(workspace), "mspi_attn_fwd_ws: workspace must be a 16-B aligned device buffer");
  return attn_fwd_impl(d, q, k, v, res, biasT, maskT, tok_idx, o, workspace, stream);
}

static int attn_fwd_impl(const MspiAttnDesc* d, const float* q, const float* k, const float* v, const float* res,
                         const float* biasT, const float* maskT, const int32_t* tok_idx, float* o, void* ws,
                         mspi_stream_t stream) {
  MSPI_REQUIRE(d && q && k && v && o, "mspi_attn_fwd: null argument");
  MSPI_REQUIRE(d->B > 0 && d->Hh > 0 && d->Nq > 0 && d->Nk > 0, "mspi_attn_fwd: empty extent");
  const int64_t st[12] = {d->q_sB, d->q_sH, d->q_sT, d->k_sB, d->k_sH, d->k_sT, d->v_sB, d->v_sH, d->v_sT, d->o_sB, d->o_sH, d->o_sT};
  for (int i = 0; i < 12; ++i) MSPI_REQUIRE((st[i] & 3) == 0, "mspi_attn_fwd: strides must be multiples of 4 floats");
  MSPI_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && (!res || aligned16(res)),
               "mspi_attn_fwd: pointers must be 16-B aligned");
  MSPI_REQUIRE((long)d->B * d->Hh < 65536, "mspi_attn_fwd: B*H too large");
  MSPI_REQUIRE(!maskT || d->nmask > 0, "mspi_attn_fwd: mask needs nmask > 0");
  MSPI_REQUIRE(!tok_idx || (d->nwin > 0 && d->B % d->nwin == 0 && d->Nq == d->Nk),
               "mspi_attn_fwd: a token index needs nwin > 0, B %% nwin == 0 and Nq == Nk");
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.res = res; a.biasT = biasT; a.maskT = maskT; a.tok_idx = tok_idx; a.o = o;
  a.B = d->B; a.Hh = d->Hh; a.Nq = d->Nq; a.Nk = d->Nk; a.nmask = d->nmask > 0 ? d->nmask : 1;
  a.nwin = d->nwin > 0 ? d->nwin : 1;
  a.kp = a.vp = nullptr; a.Nkp = 0; a.part_o = a.part_ml = nullptr;
  a.q_sB = d->q_sB; a.q_sH = d->q_sH; a.q_sT = d->q_sT;
  a.k_sB = d->k_sB; a.k_sH = d->k_sH; a.k_sT = d->k_sT;
  a.v_sB = d->v_sB; a.v_sH = d->v_sH; a.v_sT = d->v_sT;
  a.o_sB = d->o_sB; a.o_sH = d->o_sH; a.o_sT = d->o_sT;
  a.scale = d->scale;
  dim3 grid((unsigned)((d->Nq + 127) / 128), (unsigned)(d->B * d->Hh));
  hipStream_t s = (hipStream_t)stream;
  const int key = d->D * 1000 + d->Dv;
  if (d->prec == MSPI_PREC_F16X3 && ws) {
    a.Nkp = (int)attn_nkp(d);
    a.kp = reinterpret_cast<_Float16*>(ws);
    a.vp = a.kp + (size_t)d->B * d->Hh * 2 * a.Nkp * d->D;
    const size_t img_k = (size_t)d->B * d->Hh * (a.Nkp / 32) * 2 * 32 * (size_t)(d->D + 8);   // halves of all K images
    dim3 pgrid((unsigned)(a.Nkp / 32), (unsigned)(d->B * d->Hh));
    // next tile's planes fetched into registers under the current tile's MFMAs: same-box A/B on the MViTv2-S shapes 2.377 ->
    // 2.244 ms per forward, better or equal on every shape.  MSPI_ATTN_PF=0 switches it off for an A/B.
    static const char* pf_env = getenv("MSPI_ATTN_PF");
    const bool pf = !(pf_env && pf_env[0] == '0');
    // software-pipelined form (attn_pipe_kernel) wherever there is no bias, mask or token index; MSPI_ATTN_PIPE=0: off
    static const char* pipe_env = getenv("MSPI_ATTN_PIPE");
    const bool pipe = pf && !biasT && !maskT && !tok_idx && !(pipe_env && pipe_env[0] == '0');
    const int split = (pf && !biasT && !maskT && !tok_idx) ? attn_ksplit(d) : 1;
    if (split > 1) {
      a.part_o = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ws) + attn_planes_bytes(d));
      a.part_ml = a.part_o + (size_t)split * d->B * d->Hh * d->Nq * d->Dv;
      grid.z = (unsigned)split;
    }
    const dim3 mgrid((unsigned)(((long)d->B * d->Hh * d->Nq * (d->Dv / 4) + 255) / 256));
#define MSPI_ATTN_PL(DD, DVV)                                                                        \
  case DD * 1000 + DVV:                                                                              \
    if (pipe) {                                                                                      \
      a.vp = a.kp + img_k;                                                                           \
      hipLaunchKernelGGL((attn_kv_planes_kernel<DD, DVV, true>), pgrid, dim3(256), 0, s, a);         \
      hipLaunchKernelGGL((attn_pipe_kernel<DD, DVV>), grid, dim3(256), 0, s, a);                     \
      if (split > 1) hipLaunchKernelGGL((attn_merge_kernel<DVV>), mgrid, dim3(256), 0, s, a, split); \
      break;                                                                                         \
    }                                                                                                \
    hipLaunchKernelGGL((attn_kv_planes_kernel<DD, DVV>), pgrid, dim3(256), 0, s, a);                 \
    if (false) hipLaunchKernelGGL((attn_pipe_kernel<DD, DVV>), grid, dim3(256), 0, s, a);           \
    else if (pf) hipLaunchKernelGGL((attn_f16x3_kernel<DD, DVV, true, true>), grid, dim3(256), 0, s, a);  \
    else hipLaunchKernelGGL((attn_f16x3_kernel<DD, DVV, true, false>), grid, dim3(256), 0, s, a);    \
    if (split > 1) hipLaunchKernelGGL((attn_merge_kernel<DVV>), mgrid, dim3(256), 0, s, a, split);   \
    break;
    switch (key) {
      MSPI_ATTN_PL(32, 32) MSPI_ATTN_PL(64, 64) MSPI_ATTN_PL(96, 96) MSPI_ATTN_PL(128, 128) MSPI_ATTN_PL(128, 96) MSPI_ATTN_PL(144, 96) MSPI_ATTN_PL(160, 96)
      default:
        set_error("mspi_attn_fwd_ws: (D=%d, Dv=%d) not in {(32,32),(64,64),(96,96),(128,128),(128,96),(144,96),(160,96)}", d->D, d->Dv);
        return MSPI_EINVAL;
    }
#undef MSPI_ATTN_PL
    return check_launch("mspi_attn_fwd_ws");
  }
  if (d->prec == MSPI_PREC_F16X3) {
    switch (key) {
      case 32032: hipLaunchKernelGGL((attn_f16x3_kernel<32, 32>), grid, dim3(256), 0, s, a); break;
      case 64064: hipLaunchKernelGGL((attn_f16x3_kernel<64, 64>), grid, dim3(256), 0, s, a); break;
      case 96096: hipLaunchKernelGGL((attn_f16x3_kernel<96, 96>), grid, dim3(256), 0, s, a); break;
      case 128128: hipLaunchKernelGGL((attn_f16x3_kernel<128, 128>), grid, dim3(256), 0, s, a); break;
      case 128096: hipLaunchKernelGGL((attn_f16x3_kernel<128, 96>), grid, dim3(256), 0, s, a); break;
      case 144096: hipLaunchKernelGGL((attn_f16x3_kernel<144, 96>), grid, dim3(256), 0, s, a); break;
      case 160096: hipLaunchKernelGGL((attn_f16x3_kernel<160, 96>), grid, dim3(256), 0, s, a); break;
      default:
        set_error("mspi_attn_fwd: (D=%d, Dv=%d) not in {(32,32),(64,64),(96,96),(128,128),(128,96),(144,96),(160,96)}", d->D, d->Dv);
        return MSPI_EINVAL;
    }
    return check_launch("mspi_attn_fwd");
  }
  MSPI_REQUIRE(d->prec == MSPI_PREC_F32, "mspi_attn_fwd: prec = %d", d->prec);
  switch (key) {
    case 32032: hipLaunchKernelGGL((attn_kernel<32, 32>), grid, dim3(256), 0, s, a); break;
    case 64064: hipLaunchKernelGGL((attn_kernel<64, 64>), grid, dim3(256), 0, s, a); break;
    case 96096: hipLaunchKernelGGL((attn_kernel<96, 96>), grid, dim3(256), 0, s, a); break;
    case 128128: hipLaunchKernelGGL((attn_kernel<128, 128>), grid, dim3(256), 0, s, a); break;
    case 128096: hipLaunchKernelGGL((attn_kernel<128, 96>), grid, dim3(256), 0, s, a); break;
    case 144096: hipLaunchKernelGGL((attn_kernel<144, 96>), grid, dim3(256), 0, s, a); break;
    case 160096: hipLaunchKernelGGL((attn_kernel<160, 96>), grid, dim3(256), 0, s, a); break;
    default:
      set_error("mspi_attn_fwd: (D=%d, Dv=%d) not in {(32,32),(64,64),(96,96),(128,128),(128,96),(144,96),(160,96)}", d->D, d->Dv);
      return MSPI_EINVAL;
  }
  return check_launch("mspi_attn_fwd");
}
