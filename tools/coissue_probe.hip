// Micro-benchmark: do MFMA and VALU work overlap on one SIMD?
//  mode 0: 4 waves (one per SIMD) run N dependent-chain MFMAs each             -> cycles per MFMA
//  mode 1: 4 waves run N*R VALU FMAs each                                       -> cycles per VALU instruction
//  mode 2: 8 waves: waves 0-3 the MFMA loop, waves 4-7 (same SIMDs) the VALU loop  -> if overlapped: max(0, 1), else the sum
//  mode 3: 4 waves, each MFMA followed by R independent VALU FMAs in program order (same wave)
//  mode 4: 8 waves, all run mode 3's interleaved stream
//  mode 5: mode 2 with v_exp_f32 (transcendental) instead of FMA in the VALU waves
//  mode 6 / 7: mode 2 with s_setprio 3 on the MFMA waves / on the VALU waves
// One workgroup on one CU; times by s_memtime (shader clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int MODE, int R>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  v16f acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  v8h a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * e); }
  float f[8];
  for (int e = 0; e < 8; ++e) f[e] = 0.001f * (lane + e);
  const float c0 = 1.0001f, c1 = 0.0001f;
  const bool mf = (MODE == 0) || ((MODE == 2 || MODE == 6 || MODE == 7) && wave < 4) || (MODE == 5 && wave < 4) || MODE == 3 || MODE == 4;
  const bool va = (MODE == 1) || ((MODE == 2 || MODE == 6 || MODE == 7) && wave >= 4) || (MODE == 5 && wave >= 4) || MODE == 3 || MODE == 4;
  if (MODE == 6 && mf) __builtin_amdgcn_s_setprio(3);      // mode 6: mode 2 with the MFMA waves at priority 3
  if (MODE == 7 && va) __builtin_amdgcn_s_setprio(3);      // mode 7: mode 2 with the VALU waves at priority 3
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {      // 16 copies per trip: the taken branch (~40 cycles) is amortised
      if (mf) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      if (va) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (MODE == 5) f[r & 7] = __builtin_amdgcn_exp2f(f[r & 7]);
          else f[r & 7] = fmaf(f[r & 7], c0 + 0.5f * r, c1);
        }
      }
      if (mf && va) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, R, 0); }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int e = 0; e < 8; ++e) s += f[e];
  out[threadIdx.x] = s;
  if (lane == 0) cyc[wave] = t1 - t0;
}

template <int MODE, int R>
static void run(const char* what, float* out, unsigned long long* cyc, int waves) {
  const int iters = 4096;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, R>), dim3(1), dim3(waves * 64), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("mode %d R=%2d %-58s", MODE, R, what);
  for (int w = 0; w < waves; ++w) printf(" %6.1f", (double)h[w] / iters);
  printf("   (shader cycles per iteration, per wave)\n");
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 64);
  run<0, 8>("4 waves, MFMA chain", out, cyc, 4);
  run<1, 8>("4 waves, 8 FMA per iteration", out, cyc, 4);
  run<2, 8>("8 waves: 0-3 MFMA, 4-7 8 FMA", out, cyc, 8);
  run<3, 8>("4 waves: MFMA + 8 FMA interleaved in one wave", out, cyc, 4);
  run<4, 8>("8 waves: all MFMA + 8 FMA interleaved", out, cyc, 8);
  run<1, 16>("4 waves, 16 FMA per iteration", out, cyc, 4);
  run<2, 16>("8 waves: 0-3 MFMA, 4-7 16 FMA", out, cyc, 8);
  run<3, 16>("4 waves: MFMA + 16 FMA interleaved in one wave", out, cyc, 4);
  run<4, 16>("8 waves: all MFMA + 16 FMA interleaved", out, cyc, 8);
  run<1, 4>("4 waves, 4 FMA per iteration", out, cyc, 4);
  run<3, 4>("4 waves: MFMA + 4 FMA interleaved in one wave", out, cyc, 4);
  run<6, 8>("8 waves: 0-3 MFMA at s_setprio 3, 4-7 8 FMA", out, cyc, 8);
  run<7, 8>("8 waves: 0-3 MFMA, 4-7 8 FMA at s_setprio 3", out, cyc, 8);
  run<6, 16>("8 waves: 0-3 MFMA at s_setprio 3, 4-7 16 FMA", out, cyc, 8);
  run<5, 4>("8 waves: 0-3 MFMA, 4-7 4 v_exp_f32", out, cyc, 8);
  run<5, 8>("8 waves: 0-3 MFMA, 4-7 8 v_exp_f32", out, cyc, 8);
  return 0;
}
