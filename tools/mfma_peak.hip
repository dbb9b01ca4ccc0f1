// Micro-benchmark: what each ingredient of the f16x3 GEMM inner loop costs against the bare MFMA rate.
//  0 independent accumulators            1 f16x3 chain (3 MFMA per accumulator)
//  2 + 8 weight-fragment ds_read_b128    3 + activation fragment (2 ds_read_b128 fp32) and the hi/lo split (VALU)
//  4 + one __syncthreads per 2 steps     5 + 8 global_load_lds_dwordx4 per 2 steps from a 256 MB buffer (HBM misses)
//  6 same from a 4 MB window (L2 hits)   7 register staging instead: 8 global_load_dwordx4 + 8 ds_write_b128, L2 hits
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* stream, long stream_floats, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  _Float16* lh = reinterpret_cast<_Float16*>(lds);
  float* lf = reinterpret_cast<float*>(lds + 32768);
  for (int i = threadIdx.x; i < 16384; i += 256) lh[i] = (_Float16)(0.001f * (i & 63));
  for (int i = threadIdx.x; i < 8192; i += 256) lf[i] = 0.001f * (i & 127);
  __syncthreads();
  v16f acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  v8h a, al, b[4], bl[4];
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (threadIdx.x + e)); al[e] = (_Float16)0.001f; }
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) { b[j][e] = (_Float16)(0.02f * (e + j)); bl[j][e] = (_Float16)0.002f; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long sbase = (long)blockIdx.x * 4096 + threadIdx.x * 4;   // 16-B aligned; every index below is taken modulo a window < buffer
  for (int it = 0; it < iters; ++it) {
    float4 stg[8];
    if (MODE >= 5 && (it & 1) == 0) {
      const long window = MODE == 5 ? stream_floats - 4096 : 1048576;   // multiples of 4 floats, <= buffer - 16 B
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float* src = stream + (sbase + ((long)it * 8 + q) * 131072) % window;
        if (MODE == 7) stg[q] = *reinterpret_cast<const float4*>(src);
        else __builtin_amdgcn_global_load_lds(src, (lds_void*)(lds + 49152 + ((q & 3) * 4 + wave) * 1024 % 16384), 16, 0, 0);
      }
    }
    if (MODE >= 2) {
      for (int j = 0; j < 4; ++j) {
        b[j] = *reinterpret_cast<const v8h*>(&lh[((j * 32 + (lane & 31)) * 32 + (it & 3) * 8) & 8191]);
        bl[j] = *reinterpret_cast<const v8h*>(&lh[(8192 + (j * 32 + (lane & 31)) * 32 + (it & 3) * 8) & 16383]);
      }
    }
    if (MODE >= 3) {
      const float4 v0 = *reinterpret_cast<const float4*>(&lf[((wave * 32 + (lane & 31)) * 32 + ((it & 3) * 8)) & 8191]);
      const float4 v1 = *reinterpret_cast<const float4*>(&lf[((wave * 32 + (lane & 31)) * 32 + ((it & 3) * 8) + 4) & 8191]);
      const float a8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[e] = (_Float16)a8[e]; al[e] = (_Float16)(a8[e] - (float)a[e]); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[j], acc[j], 0, 0, 0);
        acc[(j + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b[j], acc[(j + 1) & 3], 0, 0, 0);
        acc[(j + 2) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl[j], acc[(j + 2) & 3], 0, 0, 0);
      } else {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b[j], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl[j], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[j], acc[j], 0, 0, 0);
      }
    }
    if (MODE == 7 && (it & 1) == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        *reinterpret_cast<float4*>(lds + 49152 + (((q & 3) * 4 + wave) * 1024 + lane * 16) % 16384) = stg[q];
    }
    if (MODE >= 4 && (it & 1)) {
      if (MODE >= 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s + lf[threadIdx.x];
}

template <int MODE>
void run(const char* name, int blocks, const float* stream, long nf) {
  float* out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, stream, nf, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, stream, nf, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 12 * 32768.0;
  printf("%-58s blocks=%4d  %8.3f ms  %7.1f TFLOP/s MFMA = %6.1f algorithmic (f16x3)\n", name, blocks, ms, flops / ms / 1e9,
         flops / ms / 1e9 / 3);
  (void)hipFree(out);
}

int main() {
  const long nf = 64L * 1024 * 1024;
  float* stream;
  (void)hipMalloc(&stream, nf * 4);
  (void)hipMemset(stream, 0, nf * 4);
  for (int blocks : {512}) {
    run<0>("0 independent accumulators", blocks, stream, nf);
    run<1>("1 f16x3 chain", blocks, stream, nf);
    run<2>("2 + 8 weight ds_read_b128 per 12 MFMA", blocks, stream, nf);
    run<3>("3 + activation fragment read + hi/lo split", blocks, stream, nf);
    run<4>("4 + barrier every 2 steps", blocks, stream, nf);
    run<5>("5 + 8 LDS-DMA (global_load_lds x4) every 2 steps, HBM", blocks, stream, nf);
    run<6>("6 + 8 LDS-DMA every 2 steps, L2-resident window", blocks, stream, nf);
    run<7>("7 + 8 global_load_dwordx4 + 8 ds_write_b128, L2 window", blocks, stream, nf);
  }
  return 0;
}
