// ISA-level probe for the packed-fp32 finding (csrc/Makefile -fno-slp-vectorize).
//
// With SLP on, hipcc compiles the bilinear coefficients of upsample_kernel to
//     v_pk_mul_f32 v[0:1], v[2:3], v[0:1] op_sel:[0,1] op_sel_hi:[0,1]
// i.e. BOTH result halves = src0.lo * src1.HI, and the destination pair IS the src1 pair: the low result reads v1, the
// high result overwrites v1.  Architecturally all sources are read before any half is written.  The observed failure
// ("low halves of a few packed results wrong, only beside MFMA-heavy kernels of another stream") is exactly what a
// hi-half-written-before-lo-half-read order would produce, so this probe runs that instruction (and two controls) in a
// loop, alone and beside an MFMA kernel on a second stream, and counts wrong halves.
//   variant 0: dst overlaps src1, cross-half op_sel (the compiler's instruction)
//   variant 1: same op_sel, dst does NOT overlap a source        (control)
//   variant 2: dst overlaps src1, plain per-half op_sel           (control)
// Build: hipcc -O2 --offload-arch=gfx950 tools/pk_overlap_probe.hip -o tools/bin/pk_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int VARIANT>
__global__ __launch_bounds__(256) void probe(unsigned long long* bad_lo, unsigned long long* bad_hi, int iters) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  unsigned long long nlo = 0, nhi = 0;
  float seed = 0.25f + 1e-3f * (float)(tid & 1023);
  for (int it = 0; it < iters; ++it) {
    // values in (0,1) like bilinear weights; distinct per lane and iteration
    v2f a, b, d;
    a.x = seed; a.y = 1.f - seed;
    b.x = 0.5f * seed + 0.125f; b.y = 0.75f - 0.5f * seed;
    const float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    if (VARIANT == 0) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "0"(b));
      nlo += (d.x != ax * by);
      nhi += (d.y != ax * by);
    } else if (VARIANT == 1) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=&v"(d) : "v"(a), "v"(b));
      nlo += (d.x != ax * by);
      nhi += (d.y != ax * by);
    } else {
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "0"(b));
      nlo += (d.x != ax * bx);
      nhi += (d.y != ay * by);
    }
    seed = seed * 0.999f + 3e-4f;
    if (seed < 0.05f) seed += 0.5f;
  }
  if (nlo) atomicAdd(bad_lo, nlo);
  if (nhi) atomicAdd(bad_hi, nhi);
}

__global__ __launch_bounds__(256, 2) void mfma_load(float* out, int iters) {
  v16f acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  v8h a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (threadIdx.x + e)); b[e] = (_Float16)(0.02f * e); }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) out[0] = s;      // keeps the loop alive
}

template <int VARIANT>
static void run(const char* what, bool beside_mfma, int iters) {
  unsigned long long *d_lo, *d_hi, h_lo = 0, h_hi = 0;
  float* sink;
  CK(hipMalloc(&d_lo, 8)); CK(hipMalloc(&d_hi, 8)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(d_lo, 0, 8)); CK(hipMemset(d_hi, 0, 8));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  const int rounds = 20;
  for (int r = 0; r < rounds; ++r) {
    if (beside_mfma) hipLaunchKernelGGL(mfma_load, dim3(512), dim3(256), 0, s1, sink, 40000);    // ~ a few ms of MFMA on every CU
    hipLaunchKernelGGL(probe<VARIANT>, dim3(1024), dim3(256), 0, s0, d_lo, d_hi, iters);
  }
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(&h_lo, d_lo, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h_hi, d_hi, 8, hipMemcpyDeviceToHost));
  const double total = (double)rounds * 1024 * 256 * iters;
  printf("%-58s %-18s wrong lo %llu, wrong hi %llu of %.3g\n", what, beside_mfma ? "beside MFMA kernel" : "alone", h_lo, h_hi, total);
  CK(hipStreamDestroy(s0)); CK(hipStreamDestroy(s1));
  CK(hipFree(d_lo)); CK(hipFree(d_hi)); CK(hipFree(sink));
}

int main() {
  const int iters = 20000;
  for (int beside = 0; beside < 2; ++beside) {
    run<0>("dst == src1, cross-half op_sel (compiler's instruction)", beside, iters);
    run<1>("dst separate, cross-half op_sel (control)", beside, iters);
    run<2>("dst == src1, per-half op_sel (control)", beside, iters);
  }
  return 0;
}
