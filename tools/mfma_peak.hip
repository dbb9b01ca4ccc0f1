// Micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate, (a) independent accumulators, (b) the f16x3 pattern
// (3 MFMAs chained on one accumulator), (c) with ds_read_b128 fragment reads in the loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (_Float16)(0.001f * (i & 63));
  __syncthreads();
  v16f acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  v8h a, al, b[4], bl[4];
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (threadIdx.x + e)); al[e] = (_Float16)0.001f; }
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) { b[j][e] = (_Float16)(0.02f * (e + j)); bl[j][e] = (_Float16)0.002f; }
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
      for (int j = 0; j < 4; ++j) {
        b[j] = *reinterpret_cast<const v8h*>(&lds[((j * 32 + (lane & 31)) * 32 + (it & 3) * 8) & 16383]);
        bl[j] = *reinterpret_cast<const v8h*>(&lds[(8192 + (j * 32 + (lane & 31)) * 32 + (it & 3) * 8) & 16383]);
      }
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
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 12 * 32768.0;
  printf("%-34s blocks=%4d  %8.3f ms  %8.1f TFLOP/s (MFMA flops)\n", name, blocks, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int blocks : {256, 512, 1024}) {
    run<0>("independent accumulators", blocks);
    run<1>("f16x3 chain (3 MFMA / acc)", blocks);
    run<2>("f16x3 chain + 8 ds_read_b128/iter", blocks);
  }
  return 0;
}
