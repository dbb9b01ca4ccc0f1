// Row-wise and element-wise HBM-bound kernels: LayerNorm (one wavefront per row),
// squeeze-excite gate, bilinear up-sample(+add), SA row gate, logsumexp, row mean,
// negative cosine, add.  All fp32, 16-B vector access along the channel axis.
#include "common.h"

namespace mspi {

// ------------------------------------------------------------------ LayerNorm
// LPR lanes per row (16 / 32 / 64, chosen so that narrow rows do not idle most of a wavefront: C = 96 is only 24
// float4), 64/LPR rows per wavefront, 4 wavefronts per workgroup.  The row stays in registers (VPT float4 per
// lane), x is read once; two-pass mean / variance like ATen's CPU kernel.  Loads are unconditional (clamped
// index + select) so they all fly together.
constexpr int LN_MAXC = 3072;

template <int LPR, int VPT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx, long sNx,
                                                        float* __restrict__ y, long ldy, long sNy,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, long M, int R, int C,
                                                        int act, const float* __restrict__ table,
                                                        _Float16* __restrict__ ys, long ldys, long yplane) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool rok = row < M;
  const long rr = rok ? row : 0;
  const long n = (long)((unsigned)rr / (unsigned)R);      // 32-bit division (M < 2^31, checked on the host): the 64-bit form is ~100 instructions
  const int r = (int)(rr - n * R);
  const int nv = C >> 2;
  const float* xr = x + n * sNx + (long)r * ldx;
  float4 v[VPT];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = sub + LPR * j;
    const float4 t = *reinterpret_cast<const float4*>(xr + (i < nv ? i : 0) * 4);
    v[j] = i < nv ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    if (sub + LPR * j < nv) {
      const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)C + eps);
  float* yr = y + n * sNy + (long)r * ldy;
  const float* tr = table ? table + (long)r * C : nullptr;
  auto fin = [&](auto act_c) {      // the activation as a compile-time constant inside the loop (common.h)
  constexpr int ACT = decltype(act_c)::value;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = sub + LPR * j;
    if (rok && i < nv) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + i * 4);
      const float4 b = *reinterpret_cast<const float4*>(beta + i * 4);
      float4 o;
      o.x = act_apply((v[j].x - mean) * rstd * g.x + b.x, ACT);
      o.y = act_apply((v[j].y - mean) * rstd * g.y + b.y, ACT);
      o.z = act_apply((v[j].z - mean) * rstd * g.z + b.z, ACT);
      o.w = act_apply((v[j].w - mean) * rstd * g.w + b.w, ACT);
      if (tr) {
        const float4 t = *reinterpret_cast<const float4*>(tr + i * 4);
        o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
      }
      if (ys) {     // pre-split output for the f16x3 GEMM that consumes these rows (mspi_gemm_sp_fwd): dense rows
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const float a[4] = {o.x, o.y, o.z, o.w};
        h4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { _Float16 hh, ll; split_f16(a[e], hh, ll); h[e] = hh; l[e] = ll; }
        const long po = plane_off(row, i * 4, C >> 5);      // blocked planes (common.h)
        *reinterpret_cast<h4*>(ys + po) = h;
        *reinterpret_cast<h4*>(ys + yplane + po) = l;
      } else {
        *reinterpret_cast<float4*>(yr + i * 4) = o;
      }
    }
  }
  };
  MSPI_DISPATCH_ACT(act, fin)
}

// ------------------------------------------------------------------ SE gate
// One workgroup per sample.  pool holds `rows` partial sums per channel (one per dwconv block, fixed order):
// they are summed here in a fixed order too (thread groups over row ranges, then a serial combine), so the
// squeeze-excite path is bitwise reproducible.  Then fc1 (C->F) by wavefront-reduced dots, ReLU, fc2, sigmoid.
// The kernel is ONE latency chain (8 workgroups on the chip): 29 launches of an X3D-L forward, 10.6 us each when the rows
// were read in batches of 8 (6 dependent batches at C = 216) and the fc weights were fetched where they are used (2 more
// round trips).  PRE (C <= 512, F <= 32): 24 rows in flight per thread (2 batches) and both weight matrices requested up front,
// into registers, so that the chain is the two row batches and the LDS phases.
template <bool PRE>
__global__ __launch_bounds__(1024) void se_gate_kernel(const float* __restrict__ pool, int rows, float inv_count,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      float* __restrict__ gate, int C, int F) {
  extern __shared__ float sm[];  // [G*C] partial sums, then [C] means, [F] hidden
  const int n = blockIdx.x;
  const int G = C <= 1024 ? 1024 / C : 1;   // row groups: 1024 threads, because the kernel is one latency chain per thread
  float* part = sm;
  float* mean = sm + G * C;
  float* hid = mean + C;
  const float* pb = pool + (long)n * rows * C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NB = PRE ? 24 : 8;          // rows in flight per thread
  float w1r[PRE ? 2 : 1][PRE ? 8 : 1];      // fc1: wave -> hidden units wave, wave + 16; lane -> channels lane + 64 j
  float w2r[PRE ? 32 : 1];                  // fc2: thread -> channel threadIdx.x (C <= 512 < 1024 threads)
  float b2r = 0.f;
  if (PRE) {
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int f = wave + 16 * fi, c = lane + 64 * j;
        w1r[fi][j] = (f < F && c < C) ? w1[(long)f * C + c] : 0.f;
      }
    const int c2 = min((int)threadIdx.x, C - 1);
#pragma unroll
    for (int f = 0; f < 32; ++f) w2r[f] = w2[(long)c2 * F + min(f, F - 1)];
    b2r = b2[c2];
  }
  // NB independent loads per thread and batch, 4 accumulators; the combine order is fixed, so still bitwise reproducible
  // (a single dependent add chain over ~700 partial rows made this tiny kernel cost 55 us).
  auto col_sum = [&](int c, int r0, int step) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < rows; r += NB * step) {
      float v[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int rr = r + u * step;
        v[u] = pb[(long)min(rr, rows - 1) * C + c];
        if (rr >= rows) v[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) a[u & 3] += v[u];
    }
    return (a[0] + a[1]) + (a[2] + a[3]);
  };
  if (G > 1) {
    const int c = threadIdx.x % C, g = threadIdx.x / C;
    if (g < G) part[g * C + c] = col_sum(c, g, G);
  } else {
    for (int c = threadIdx.x; c < C; c += 1024) part[c] = col_sum(c, 0, 1);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 1024) {
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += part[g * C + c];
    mean[c] = s * inv_count;
  }
  __syncthreads();
  if (PRE) {
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
      const int f = wave + 16 * fi;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j;
        s = fmaf(w1r[fi][j], c < C ? mean[c] : 0.f, s);
      }
      s = wave_sum(s);
      if (lane == 0 && f < F) hid[f] = fmaxf(s + b1[f], 0.f);
    }
  } else {
    for (int f = wave; f < F; f += 16) {
      float s = 0.f;
      for (int c = lane; c < C; c += 64) s = fmaf(w1[(long)f * C + c], mean[c], s);
      s = wave_sum(s);
      if (lane == 0) hid[f] = fmaxf(s + b1[f], 0.f);
    }
  }
  __syncthreads();
  if (PRE) {
    if ((int)threadIdx.x < C) {
      float s = b2r;
#pragma unroll
      for (int f = 0; f < 32; ++f) s = fmaf(w2r[f], f < F ? hid[f] : 0.f, s);
      gate[(long)n * C + threadIdx.x] = fast_sigmoid(s);
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 1024) {
      float s = b2[c];
      for (int f = 0; f < F; ++f) s = fmaf(w2[(long)c * F + f], hid[f], s);
      gate[(long)n * C + c] = fast_sigmoid(s);
    }
  }
}

// ------------------------------------------------------------------ bilinear up-sample (+ add)
// PyTorch align_corners=False: src = max(0, (dst + 0.5)/k - 0.5); i1 = min(i0 + 1, in - 1).
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst,
                                                       long ldd, int NT, int H, int W, int CV, int k, int accumulate,
                                                       int act) {
  const int Ho = H * k, Wo = W * k;
  const long total = (long)NT * Ho * Wo * CV;
  const float inv = 1.f / (float)k;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    // 32-bit index arithmetic (total < 2^31, checked on the host): four 64-bit divisions per output vector otherwise
    const unsigned ui = (unsigned)idx;
    const int cv = (int)(ui % (unsigned)CV);
    unsigned pos = ui / (unsigned)CV;
    const int wo = (int)(pos % (unsigned)Wo);
    pos /= (unsigned)Wo;
    const int ho = (int)(pos % (unsigned)Ho);
    const long nt = (long)(pos / (unsigned)Ho);
    float fh = ((float)ho + 0.5f) * inv - 0.5f;
    float fw = ((float)wo + 0.5f) * inv - 0.5f;
    fh = fh < 0.f ? 0.f : fh;
    fw = fw < 0.f ? 0.f : fw;
    const int h0 = (int)fh, w0 = (int)fw;
    const int h1 = h0 + (h0 < H - 1 ? 1 : 0), w1 = w0 + (w0 < W - 1 ? 1 : 0);
    const float lh = fh - (float)h0, lw = fw - (float)w0;
    const float* b = src + (nt * H * W) * lds + cv * 4;
    const float4 v00 = *reinterpret_cast<const float4*>(b + ((long)h0 * W + w0) * lds);
    const float4 v01 = *reinterpret_cast<const float4*>(b + ((long)h0 * W + w1) * lds);
    const float4 v10 = *reinterpret_cast<const float4*>(b + ((long)h1 * W + w0) * lds);
    const float4 v11 = *reinterpret_cast<const float4*>(b + ((long)h1 * W + w1) * lds);
    const float c00 = (1.f - lh) * (1.f - lw), c01 = (1.f - lh) * lw, c10 = lh * (1.f - lw), c11 = lh * lw;
    float4 o;
    o.x = c00 * v00.x + c01 * v01.x + c10 * v10.x + c11 * v11.x;
    o.y = c00 * v00.y + c01 * v01.y + c10 * v10.y + c11 * v11.y;
    o.z = c00 * v00.z + c01 * v01.z + c10 * v10.z + c11 * v11.z;
    o.w = c00 * v00.w + c01 * v01.w + c10 * v10.w + c11 * v11.w;
    float* d = dst + ((nt * Ho + ho) * Wo + wo) * ldd + cv * 4;
    if (accumulate) {
      const float4 p = *reinterpret_cast<const float4*>(d);
      o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
    }
    if (act != MSPI_ACT_NONE) {
      auto fin = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
        o.x = act_apply(o.x, ACT); o.y = act_apply(o.y, ACT); o.z = act_apply(o.z, ACT); o.w = act_apply(o.w, ACT);
      };
      MSPI_DISPATCH_ACT(act, fin)
    }
    *reinterpret_cast<float4*>(d) = o;
  }
}

// ------------------------------------------------------------------ PatchMerging gather (2x2 space-to-depth)
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy,
                                                  int NT, int H, int W, int CV) {
  const int Ho = H / 2, Wo = W / 2;
  const long total = (long)NT * Ho * Wo * 4 * CV;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cv = (int)(idx % CV);
    long r = idx / CV;
    const int qd = (int)(r & 3);
    r >>= 2;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const long nt = r / Ho;
    const int dh = qd & 1, dw = qd >> 1;   // q = 0:(0,0) 1:(1,0) 2:(0,1) 3:(1,1)
    const float4 v = *reinterpret_cast<const float4*>(x + ((nt * H + 2 * ho + dh) * W + 2 * wo + dw) * ldx + cv * 4);
    *reinterpret_cast<float4*>(y + ((nt * Ho + ho) * Wo + wo) * ldy + (long)(qd * CV + cv) * 4) = v;
  }
}

// ------------------------------------------------------------------ SA gate  x *= (1 + mask[row])
__global__ __launch_bounds__(256) void rowgate_kernel(float* __restrict__ x, long ldx, const float* __restrict__ mask,
                                                      long M, int CV) {
  const long total = M * CV;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long row = idx / CV;
    const int cv = (int)(idx - row * CV);
    const float g = 1.f + mask[row];
    float4* p = reinterpret_cast<float4*>(x + row * ldx + cv * 4);
    float4 v = *p;
    v.x *= g; v.y *= g; v.z *= g; v.w *= g;
    *p = v;
  }
}

// ------------------------------------------------------------------ block reductions
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  return r;
}

// x[n,:] -= logsumexp(x[n,:]); one 1024-thread workgroup per sample (L = H*W = 50k floats, L2 resident)
__global__ __launch_bounds__(1024) void logsumexp_sub_kernel(float* __restrict__ x, int L) {
  __shared__ float red[16];
  float* xr = x + (long)blockIdx.x * L;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < L; i += 1024) m = fmaxf(m, xr[i]);
  m = block_reduce(m, red, true);
  float s = 0.f;
  for (int i = threadIdx.x; i < L; i += 1024) s += expf(xr[i] - m);
  s = block_reduce(s, red, false);
  const float lse = m + logf(s);
  for (int i = threadIdx.x; i < L; i += 1024) xr[i] -= lse;
}

// out[n,c] = mean_r x[n,r,c]; grid (ceil(C/64), N), 256 threads = 64 channels x 4 row groups
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ x, long ldx, long sample_stride,
                                                        float* __restrict__ out, int R, int C) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  const int n = blockIdx.y;
  float s = 0.f;
  if (c < C) {
    const float* b = x + (long)n * sample_stride + c;
    for (int r = g; r < R; r += 4) s += b[(long)r * ldx];
  }
  part[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && c < C) out[(long)n * C + c] = (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) / (float)R;
}

// Two-stage form for long samples (MorphMLP's re-weighting pools 25088 rows): stage 1 sums `chunk` rows per block into
// ws[n][slice][C] (grid (ceil(C/64), slices, N)), stage 2 adds the slices in fixed order and scales by 1/R.  No atomics.
__global__ __launch_bounds__(256) void sum_rows_slice_kernel(const float* __restrict__ x, long ldx, long sample_stride,
                                                             float* __restrict__ ws, int R, int C, int chunk) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  const int n = blockIdx.z;
  const int r0 = blockIdx.y * chunk, r1 = min(R, r0 + chunk);
  float s = 0.f;
  if (c < C) {
    const float* b = x + (long)n * sample_stride + c;
    for (int r = r0 + g; r < r1; r += 4) s += b[(long)r * ldx];
  }
  part[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && c < C)
    ws[((long)n * gridDim.y + blockIdx.y) * C + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void sum_slices_kernel(const float* __restrict__ ws, float* __restrict__ out, int S, int C,
                                                         float inv, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long n = idx / C;
  const int c = (int)(idx - n * C);
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += ws[(n * S + k) * C + c];
  out[idx] = s * inv;
}

// out (+)= scale * mean_n( -cos(p_n, z_n) ), F.cosine_similarity eps = 1e-8 on each norm
__global__ __launch_bounds__(256) void neg_cosine_kernel(const float* __restrict__ p, const float* __restrict__ z,
                                                         float* __restrict__ out, int N, int C, float scale,
                                                         int accumulate) {
  __shared__ float red[4];
  float total = 0.f;
  for (int n = 0; n < N; ++n) {
    float dot = 0.f, pp = 0.f, zz = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
      const float a = p[(long)n * C + c], b = z[(long)n * C + c];
      dot = fmaf(a, b, dot);
      pp = fmaf(a, a, pp);
      zz = fmaf(b, b, zz);
    }
    dot = block_reduce(dot, red, false);
    pp = block_reduce(pp, red, false);
    zz = block_reduce(zz, red, false);
    total += dot / (fmaxf(sqrtf(pp), 1e-8f) * fmaxf(sqrtf(zz), 1e-8f));
  }
  if (threadIdx.x == 0) {
    const float v = -scale * total / (float)N;
    out[0] = accumulate ? out[0] + v : v;
  }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long n4, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(y)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
  if (blockIdx.x == 0) {
    const long i = n4 * 4 + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
  }
}

static inline unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  const long cap = 256L * 16;  // 256 CUs x 16 resident 256-thread workgroups, grid-stride beyond that
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace mspi

using namespace mspi;

static int layernorm_impl(const float* x, int64_t ldx, int64_t sNx, float* y, int64_t ldy, int64_t sNy, void* planes, int64_t ldo,
                          int64_t plane, const float* gamma, const float* beta, float eps, int32_t N, int32_t R, int32_t C,
                          int32_t act, const float* table, mspi_stream_t stream, const char* who) {
  MSPI_REQUIRE(x && (y || planes) && gamma && beta, "%s: null argument", who);
  const int64_t M = (int64_t)N * R;
  MSPI_REQUIRE(M < (1L << 31), "%s: more than 2^31 rows", who);
  MSPI_REQUIRE(N > 0 && R > 0 && C > 0 && (C & 3) == 0 && C <= LN_MAXC, "%s: C=%d must be a multiple of 4, <= %d", who, C, LN_MAXC);
  MSPI_REQUIRE((ldx & 3) == 0 && ldx >= C && (sNx & 3) == 0, "%s: bad input row/sample stride", who);
  MSPI_REQUIRE(planes || ((ldy & 3) == 0 && ldy >= C && (sNy & 3) == 0 && aligned16(y)), "%s: bad output row/sample stride", who);
  MSPI_REQUIRE(!planes || ((C & 31) == 0 && ldo == C && plane >= (M + 15) / 16 * 16 * ldo && (plane & 7) == 0 && aligned16(planes)),
               "%s: blocked output planes need C %% 32 == 0, ld == C and plane >= roundup16(M)*C", who);
  MSPI_REQUIRE(aligned16(x) && aligned16(gamma) && aligned16(beta) && (!table || aligned16(table)), "%s: pointers must be 16-B aligned", who);
  MSPI_REQUIRE(M < (1L << 31), "%s: too many rows", who);
  const int nv = C / 4;
  hipStream_t s = (hipStream_t)stream;
#define MSPI_LN(LPR, VPT)                                                                                             \
  hipLaunchKernelGGL((layernorm_kernel<LPR, VPT>), dim3((unsigned)((M + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)))), dim3(256), \
                     0, s, x, (long)ldx, (long)sNx, y, (long)ldy, (long)sNy, gamma, beta, eps, (long)M, R, C, act, table,    \
                     (_Float16*)planes, (long)ldo, (long)plane)
  if (nv <= 16) MSPI_LN(16, 1);
  else if (nv <= 32) MSPI_LN(16, 2);
  else if (nv <= 64) MSPI_LN(16, 4);
  else if (nv <= 128) MSPI_LN(32, 4);
  else if (nv <= 256) MSPI_LN(64, 4);
  else if (nv <= 512) MSPI_LN(64, 8);
  else MSPI_LN(64, 12);
#undef MSPI_LN
  return check_launch(who);
}

extern "C" int mspi_layernorm_fwd(const float* x, int64_t ldx, int64_t sNx, float* y, int64_t ldy, int64_t sNy,
                                  const float* gamma, const float* beta, float eps, int32_t N, int32_t R, int32_t C,
                                  int32_t act, const float* table, mspi_stream_t stream) {
  return layernorm_impl(x, ldx, sNx, y, ldy, sNy, nullptr, 0, 0, gamma, beta, eps, N, R, C, act, table, stream, "mspi_layernorm_fwd");
}

extern "C" int mspi_layernorm_sp_fwd(const float* x, int64_t ldx, int64_t sNx, void* planes, int64_t ldo, int64_t plane,
                                     const float* gamma, const float* beta, float eps, int32_t N, int32_t R, int32_t C,
                                     int32_t act, mspi_stream_t stream) {
  MSPI_REQUIRE(planes, "mspi_layernorm_sp_fwd: null output planes");
  return layernorm_impl(x, ldx, sNx, nullptr, 0, 0, planes, ldo, plane, gamma, beta, eps, N, R, C, act, nullptr, stream,
                        "mspi_layernorm_sp_fwd");
}

extern "C" int mspi_se_gate(const float* pool, int32_t rows, float inv_count, const float* w1, const float* b1,
                            const float* w2, const float* b2, float* gate, int32_t N, int32_t C, int32_t F,
                            mspi_stream_t stream) {
  MSPI_REQUIRE(pool && w1 && b1 && w2 && b2 && gate, "mspi_se_gate: null argument");
  const int G = C <= 1024 ? 1024 / C : 1;
  const size_t lds = (size_t)(G * C + C + F) * sizeof(float);
  MSPI_REQUIRE(N > 0 && rows > 0 && C > 0 && F > 0 && lds <= 64 * 1024, "mspi_se_gate: bad extent");
  if (C <= 512 && F <= 32) hipLaunchKernelGGL((se_gate_kernel<true>), dim3(N), dim3(1024), lds, (hipStream_t)stream, pool, rows, inv_count, w1, b1, w2,
                     b2, gate, C, F);
  else hipLaunchKernelGGL((se_gate_kernel<false>), dim3(N), dim3(1024), lds, (hipStream_t)stream, pool, rows, inv_count, w1, b1, w2,
                     b2, gate, C, F);
  return check_launch("mspi_se_gate");
}

extern "C" int mspi_upsample_fwd(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t NT, int32_t H,
                                 int32_t W, int32_t C, int32_t factor, int32_t accumulate, int32_t act,
                                 mspi_stream_t stream) {
  MSPI_REQUIRE(src && dst, "mspi_upsample_fwd: null argument");
  MSPI_REQUIRE(NT > 0 && H > 0 && W > 0 && C > 0 && factor >= 1, "mspi_upsample_fwd: bad extent");
  MSPI_REQUIRE((C & 3) == 0 && (lds & 3) == 0 && (ldd & 3) == 0 && lds >= C && ldd >= C && aligned16(src) && aligned16(dst),
               "mspi_upsample_fwd: C/ld must be multiples of 4, pointers 16-B aligned");
  const long total = (long)NT * H * factor * W * factor * (C / 4);
  MSPI_REQUIRE(total < (1L << 31), "mspi_upsample_fwd: more than 2^31 output vectors");
  hipLaunchKernelGGL(upsample_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, (long)lds, dst,
                     (long)ldd, NT, H, W, C / 4, factor, accumulate, act);
  return check_launch("mspi_upsample_fwd");
}

extern "C" int mspi_space_to_depth(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t NT, int32_t H, int32_t W,
                                   int32_t C, mspi_stream_t stream) {
  MSPI_REQUIRE(x && y && NT > 0 && H > 0 && W > 0 && C > 0, "mspi_space_to_depth: bad argument");
  MSPI_REQUIRE((H & 1) == 0 && (W & 1) == 0, "mspi_space_to_depth: H and W must be even (PatchMerging pads odd sizes; not supported)");
  MSPI_REQUIRE((C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= C && ldy >= 4 * C && aligned16(x) && aligned16(y),
               "mspi_space_to_depth: C/ld multiples of 4, 16-B alignment");
  hipLaunchKernelGGL(s2d_kernel, dim3(grid_for((long)NT * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, y,
                     (long)ldy, NT, H, W, C / 4);
  return check_launch("mspi_space_to_depth");
}

extern "C" int mspi_rowgate(float* x, int64_t ldx, const float* mask, int64_t M, int32_t C, mspi_stream_t stream) {
  MSPI_REQUIRE(x && mask && M > 0 && C > 0, "mspi_rowgate: bad argument");
  MSPI_REQUIRE((C & 3) == 0 && (ldx & 3) == 0 && ldx >= C && aligned16(x), "mspi_rowgate: C/ldx must be multiples of 4");
  hipLaunchKernelGGL(rowgate_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, mask,
                     (long)M, C / 4);
  return check_launch("mspi_rowgate");
}

extern "C" int mspi_logsumexp_sub(float* x, int32_t N, int32_t L, mspi_stream_t stream) {
  MSPI_REQUIRE(x && N > 0 && L > 0, "mspi_logsumexp_sub: bad argument");
  hipLaunchKernelGGL(logsumexp_sub_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, x, L);
  return check_launch("mspi_logsumexp_sub");
}

extern "C" int mspi_mean_rows(const float* x, int64_t ldx, int64_t sample_stride, float* out, int32_t N, int32_t R,
                              int32_t C, mspi_stream_t stream) {
  MSPI_REQUIRE(x && out && N > 0 && R > 0 && C > 0 && N < 65536, "mspi_mean_rows: bad argument");
  hipLaunchKernelGGL(mean_rows_kernel, dim3((C + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, x, (long)ldx,
                     (long)sample_stride, out, R, C);
  return check_launch("mspi_mean_rows");
}

extern "C" int mspi_mean_rows_slices(int32_t R) { return R >= 1024 ? (R + 255) / 256 : 0; }

extern "C" int mspi_mean_rows_ws(const float* x, int64_t ldx, int64_t sample_stride, float* out, float* ws, int32_t N,
                                 int32_t R, int32_t C, mspi_stream_t stream) {
  MSPI_REQUIRE(x && out && ws && N > 0 && R > 0 && C > 0 && N < 65536, "mspi_mean_rows_ws: bad argument");
  const int S = mspi_mean_rows_slices(R);
  MSPI_REQUIRE(S > 0 && S < 65536, "mspi_mean_rows_ws: R=%d is served by mspi_mean_rows (no workspace)", R);
  hipLaunchKernelGGL(sum_rows_slice_kernel, dim3((C + 63) / 64, S, N), dim3(256), 0, (hipStream_t)stream, x, (long)ldx,
                     (long)sample_stride, ws, R, C, 256);
  const long total = (long)N * C;
  hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, out, S, C,
                     1.f / (float)R, total);
  return check_launch("mspi_mean_rows_ws");
}

extern "C" int mspi_neg_cosine(const float* p, const float* z, float* out, int32_t N, int32_t C, float scale,
                               int32_t accumulate, mspi_stream_t stream) {
  MSPI_REQUIRE(p && z && out && N > 0 && C > 0, "mspi_neg_cosine: bad argument");
  hipLaunchKernelGGL(neg_cosine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, z, out, N, C, scale, accumulate);
  return check_launch("mspi_neg_cosine");
}

extern "C" int mspi_add(const float* a, const float* b, float* y, int64_t n, mspi_stream_t stream) {
  MSPI_REQUIRE(a && b && y && n > 0, "mspi_add: bad argument");
  MSPI_REQUIRE(aligned16(a) && aligned16(b) && aligned16(y), "mspi_add: pointers must be 16-B aligned");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, b, y, (long)(n / 4), (long)n);
  return check_launch("mspi_add");
}
