// LDS read forms on gfx950: ds_read_b64 x2 vs ds_read2_b64 vs ds_read2st64_b64 vs ds_read_b128, 12 waves per CU, all CUs.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds_read2.hip -o scripts/ubench_lds_read2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(768) void k(float* out, int iters) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 12 * 2304; i += 768) smem[i] = (float)i;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(smem + wave * 2304) + lane * 8;   // conflict-free 8-byte units
  v2f acc = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {   // 16 x ds_read_b64 (two per "pair")
      v2f a[16];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[2 * t]) : "v"(base), "n"(t * 1024));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[2 * t + 1]) : "v"(base), "n"(t * 1024 + 512));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 16; ++t) acc += a[t];
    } else if (MODE == 1) {   // 8 x ds_read2_b64 (same 16 values)
      v4f a[8];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(a[t]) : "v"(base + (t >> 1) * 2048), "n"((t & 1) * 128), "n"((t & 1) * 128 + 64));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += v2f{a[t].x + a[t].z, a[t].y + a[t].w};
    } else if (MODE == 2) {   // 8 x ds_read2st64_b64
      v4f a[8];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(a[t]) : "v"(base), "n"(2 * t), "n"(2 * t + 1));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += v2f{a[t].x + a[t].z, a[t].y + a[t].w};
    } else {   // 8 x ds_read_b128 (16 bytes per lane, contiguous)
      v4f a[8];
      const unsigned b16 = (unsigned)(size_t)(smem + wave * 2304) + lane * 16;
#pragma unroll
      for (int t = 0; t < 8; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[t]) : "v"(b16), "n"(t * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += v2f{a[t].x + a[t].z, a[t].y + a[t].w};
    }
  }
  out[blockIdx.x * 768 + threadIdx.x] = acc.x + acc.y;
}
template <int MODE>
float run(float* d, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 2304 * 4);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 12 * 2304 * 4, 0, d, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 12 * 2304 * 4, 0, d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 768 * 4);
  const int iters = 20000;
  const char* names[4] = {"16 x ds_read_b64", "8 x ds_read2_b64", "8 x ds_read2st64_b64", "8 x ds_read_b128"};
  float ms[4] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters)};
  for (int m = 0; m < 4; ++m) {
    // bytes per CU per iteration: 12 waves x 64 lanes x 128 B
    const double cyc = ms[m] * 1e-3 * 2.4e9 / iters;
    printf("%-22s %.3f ms  -> %.1f cycles (at 2.4 GHz) per 12-wave round of 128 B/lane = %.0f B/clk/CU\n", names[m], ms[m], cyc, 12 * 64 * 128 / cyc);
  }
  return 0;
}
