// Micro-benchmark: issue cost of one 1-KB LDS-DMA instruction (global_load_lds, 16 B per lane) by source pattern, 8 waves per CU on
// all 256 CUs, L2-resident source window.  pattern 0: 1 KB contiguous (8 full 128-B lines); 1: 16 rows x 64 B, row pitch PITCH
// (16 half lines: what a k32 stage of f16 planes [M][K] is); 2: 8 rows x 128 B, row pitch PITCH (8 full lines: k64 stage).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;

template <int PAT>
__global__ __launch_bounds__(512, 1) void k(const unsigned char* src, long window, int pitch, int iters, unsigned long long* cyc, float* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long off;
  if (PAT == 0) off = (long)lane * 16;
  else if (PAT == 1) off = (long)(lane >> 2) * pitch + (lane & 3) * 16;
  else off = (long)(lane >> 3) * pitch + (lane & 7) * 16;
  const long base = ((long)blockIdx.x * 8 + wave) * 65536 % window;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long o = (base + ((long)it * 8 + u) * (PAT == 0 ? 1024 : PAT == 1 ? 16L * pitch : 8L * pitch)) % window;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + o + off), (lds_void*)(lds + (wave * 8 + u) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  out[blockIdx.x * 512 + threadIdx.x] = lds[threadIdx.x * 16];
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

int main() {
  const long window = 8L << 20;      // 8 MB: L2 / MALL resident
  unsigned char* src; unsigned long long* cyc; float* out;
  hipMalloc(&src, window + (1 << 20)); hipMemset(src, 1, window + (1 << 20));
  hipMalloc(&cyc, 64); hipMalloc(&out, 256 * 512 * 4);
  const int iters = 256;
  for (int pat = 0; pat < 3; ++pat)
    for (int pitch : {768, 3072}) {
      if (pat == 0 && pitch != 768) continue;
      for (int rep = 0; rep < 2; ++rep) {
        if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, src, window, pitch, iters, cyc, out);
        if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, src, window, pitch, iters, cyc, out);
        if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, src, window, pitch, iters, cyc, out);
      }
      hipDeviceSynchronize();
      unsigned long long h[8];
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double m = 0; for (int w = 0; w < 8; ++w) m += (double)h[w] / (iters * 8);
      printf("pattern %d pitch %4d: %.1f shader cycles per 1-KB DMA instruction per wave (8 waves per CU issuing)  -> %.1f B/clk/CU\n",
             pat, pitch, m / 8, 8 * 1024.0 / (m / 8));
    }
  return 0;
}
