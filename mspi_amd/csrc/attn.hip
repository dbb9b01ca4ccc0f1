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
  float* o;
  int B, Hh, Nq, Nk;
  long q_sB, q_sH, q_sT, k_sB, k_sH, k_sT, v_sB, v_sH, v_sT, o_sB, o_sH, o_sT;
  float scale;
};

template <int D>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs p) {
  constexpr int LDD = D + 4;
  constexpr int NC = D / 8;   // float4 chunks of one lane half
  constexpr int NT = D / 32;  // 32-wide output tiles along d
  __shared__ __attribute__((aligned(16))) float smem[2 * 32 * LDD];
  float* Ks = smem;
  float* Vs = smem + 32 * LDD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y / p.Hh, h = blockIdx.y % p.Hh;
  const int q = blockIdx.x * 128 + wave * 32 + li;
  const bool qok = q < p.Nq;

  float4 qr[NC];
  {
    const float* qp = p.q + (long)b * p.q_sB + (long)h * p.q_sH + (long)(qok ? q : 0) * p.q_sT;
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

  const float* kb = p.k + (long)b * p.k_sB + (long)h * p.k_sH;
  const float* vb = p.v + (long)b * p.v_sB + (long)h * p.v_sH;

  for (int k0 = 0; k0 < p.Nk; k0 += 32) {
    __syncthreads();  // previous tile fully consumed
    for (int idx = tid; idx < 32 * (D / 4); idx += 256) {
      const int row = idx / (D / 4), c4 = idx - row * (D / 4);
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (k0 + row < p.Nk) {
        kv = *reinterpret_cast<const float4*>(kb + (long)(k0 + row) * p.k_sT + c4 * 4);
        vv = *reinterpret_cast<const float4*>(vb + (long)(k0 + row) * p.v_sT + c4 * 4);
      }
      *reinterpret_cast<float4*>(&Ks[row * LDD + c4 * 4]) = kv;
      *reinterpret_cast<float4*>(&Vs[row * LDD + c4 * 4]) = vv;
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
        const float vf = Vs[krow * LDD + t * 32 + li];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], acc[t], 0, 0, 0);
      }
    }
  }

  if (qok) {
    const float inv = 1.f / l_run;
    float* op = p.o + (long)b * p.o_sB + (long)h * p.o_sH + (long)q * p.o_sT;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // registers 4g..4g+3 are 4 consecutive d: d = t*32 + 8g + 4*lh + (0..3)
        const float4 o4 = make_float4(acc[t][4 * g] * inv, acc[t][4 * g + 1] * inv, acc[t][4 * g + 2] * inv,
                                      acc[t][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + t * 32 + 8 * g + 4 * lh) = o4;
      }
  }
}

}  // namespace mspi

using namespace mspi;

extern "C" int mspi_attn_fwd(const MspiAttnDesc* d, const float* q, const float* k, const float* v, float* o,
                             mspi_stream_t stream) {
  MSPI_REQUIRE(d && q && k && v && o, "mspi_attn_fwd: null argument");
  MSPI_REQUIRE(d->B > 0 && d->Hh > 0 && d->Nq > 0 && d->Nk > 0, "mspi_attn_fwd: empty extent");
  MSPI_REQUIRE(d->D == 32 || d->D == 64 || d->D == 96 || d->D == 128, "mspi_attn_fwd: head_dim %d not in {32,64,96,128}", d->D);
  const int64_t st[12] = {d->q_sB, d->q_sH, d->q_sT, d->k_sB, d->k_sH, d->k_sT, d->v_sB, d->v_sH, d->v_sT, d->o_sB, d->o_sH, d->o_sT};
  for (int i = 0; i < 12; ++i) MSPI_REQUIRE((st[i] & 3) == 0, "mspi_attn_fwd: strides must be multiples of 4 floats");
  MSPI_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o), "mspi_attn_fwd: pointers must be 16-B aligned");
  MSPI_REQUIRE((long)d->B * d->Hh < 65536, "mspi_attn_fwd: B*H too large");
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.o = o;
  a.B = d->B; a.Hh = d->Hh; a.Nq = d->Nq; a.Nk = d->Nk;
  a.q_sB = d->q_sB; a.q_sH = d->q_sH; a.q_sT = d->q_sT;
  a.k_sB = d->k_sB; a.k_sH = d->k_sH; a.k_sT = d->k_sT;
  a.v_sB = d->v_sB; a.v_sH = d->v_sH; a.v_sT = d->v_sT;
  a.o_sB = d->o_sB; a.o_sH = d->o_sH; a.o_sT = d->o_sT;
  a.scale = d->scale;
  dim3 grid((unsigned)((d->Nq + 127) / 128), (unsigned)(d->B * d->Hh));
  hipStream_t s = (hipStream_t)stream;
  switch (d->D) {
    case 32: hipLaunchKernelGGL((attn_kernel<32>), grid, dim3(256), 0, s, a); break;
    case 64: hipLaunchKernelGGL((attn_kernel<64>), grid, dim3(256), 0, s, a); break;
    case 96: hipLaunchKernelGGL((attn_kernel<96>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((attn_kernel<128>), grid, dim3(256), 0, s, a); break;
  }
  return check_launch("mspi_attn_fwd");
}
