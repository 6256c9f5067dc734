// Matrix-pipe issue rates on gfx950 in the conv kernels' instruction patterns, all CUs.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma.hip -o scripts/ubench_mfma.bin
// MODE 0: fp32 16x16x4, 10 independent accumulators (5 A x 2 B registers), nothing else
// MODE 1: the same with 7 ds_read_b32 per 10 MFMAs interleaved one per MFMA (conv1_resident_kernel's k-step)
// MODE 2: f16 16x16x32, 10 accumulators x 3 dependent terms, nothing else
// MODE 3: the same with 14 ds_read_b128 in front of every 30 MFMAs (conv1_f16e_kernel<3>'s k-step)
// MODE 6 / 7: mode 1 + 2 / 10 independent v_fma_f32 per k-step of 10 MFMAs (what VALU work inside a k-step costs the pipe)
// MODE 4 / 5: modes 2 / 0 on PSEUDO-RANDOM operands (every lane, register and half different): the power of real data
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k(float* out, int iters) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < NW * 2304; i += NW * 64) {
    if (MODE >= 4) {
      unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
      h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
      if (MODE == 4) {   // two float16 in [-1, 1)
        const _Float16 lo = (_Float16)((float)(h & 0xffff) / 32768.f - 1.f), hi = (_Float16)((float)(h >> 16) / 32768.f - 1.f);
        unsigned short a, b;
        __builtin_memcpy(&a, &lo, 2), __builtin_memcpy(&b, &hi, 2);
        const unsigned w = a | ((unsigned)b << 16);
        __builtin_memcpy(&smem[i], &w, 4);
      } else smem[i] = (float)(h >> 8) / 8388608.f - 1.f;
    } else smem[i] = (float)(i & 7) * 0.125f;
  }
  __syncthreads();
  f32x4 acc[5][2];
  for (int t = 0; t < 5; ++t)
    for (int n = 0; n < 2; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (MODE <= 1 || MODE >= 5) {
    const float* pb = smem + wave * 2304 + lane;
    float a[2][5], b[2][2], vx[10], vw = 1.0001f;
    for (int q = 0; q < 10; ++q) vx[q] = (float)(lane + q);
    for (int t = 0; t < 5; ++t) a[0][t] = pb[64 * t], a[1][t] = pb[64 * (t + 8)];
    for (int n = 0; n < 2; ++n) b[0][n] = pb[64 * (5 + n)], b[1][n] = pb[64 * (13 + n)];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 14; ++ks) {
        const int cu = ks & 1, nx = cu ^ 1;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          acc[i / 2][i % 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cu][i / 2], b[cu][i % 2], acc[i / 2][i % 2], 0, 0, 0);
          if (MODE == 1 || MODE >= 6) {
            if (i < 2) b[nx][i] = pb[64 * (5 + i) + ks];
            else if (i < 7) a[nx][i - 2] = pb[64 * (i - 2) + ks];
          }
          if ((MODE == 6 && i >= 8) || MODE == 7) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(vx[i]) : "v"(vw));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
    const h16x8* pb = reinterpret_cast<const h16x8*>(smem + wave * 2304) + lane;
    h16x8 ah[2][5], al[2][5], bh[2][2], bl[2][2];
    for (int t = 0; t < 5; ++t) ah[0][t] = pb[64 * t], ah[1][t] = pb[64 * t + 320], al[0][t] = pb[64 * t + 1], al[1][t] = pb[64 * t + 321];
    for (int n = 0; n < 2; ++n) bh[0][n] = pb[64 * n + 2], bh[1][n] = pb[64 * n + 322], bl[0][n] = pb[64 * n + 3], bl[1][n] = pb[64 * n + 323];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 14; ++ks) {
        const int cu = ks & 1, nx = cu ^ 1;
        if (MODE == 3) {
          for (int n = 0; n < 2; ++n) bh[nx][n] = pb[64 * n + (ks & 1)], bl[nx][n] = pb[64 * (2 + n) + (ks & 1)];
          for (int t = 0; t < 5; ++t) ah[nx][t] = pb[2 * t + (ks & 1)], al[nx][t] = pb[2 * t + 1 + (ks & 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cu][t], bh[cu][n], acc[t][n], 0, 0, 0);
            acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cu][t], bl[cu][n], acc[t][n], 0, 0, 0);
            acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cu][t], bh[cu][n], acc[t][n], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
  if (MODE >= 6) { float* o2 = out + 256 * 1024 - 64; }
  for (int t = 0; t < 5; ++t)
    for (int n = 0; n < 2; ++n) s += acc[t][n][0] + acc[t][n][1] + acc[t][n][2] + acc[t][n][3];
  out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}
template <int MODE, int NW>
void run(float* d, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, NW * 2304 * 4);
  hipLaunchKernelGGL((k<MODE, NW>), dim3(256), dim3(NW * 64), NW * 2304 * 4, 0, d, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, NW>), dim3(256), dim3(NW * 64), NW * 2304 * 4, 0, d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double n_mfma = (double)iters * 14 * ((MODE <= 1 || MODE >= 5) ? 10 : 30) * NW * 256;
  const double flops = n_mfma * ((MODE <= 1 || MODE >= 5) ? 16 * 16 * 4 * 2 : 16 * 16 * 32 * 2);
  const double cyc_per = ms * 1e-3 * 2.4e9 / ((double)iters * 14 * ((MODE <= 1 || MODE >= 5) ? 10 : 30) * (NW / 4));
  printf("mode %d  %2d waves/CU  %.3f ms  %.1f TFLOP/s  %.2f cycles (at 2.4 GHz) per MFMA per SIMD\n", MODE, NW, ms, flops / ms * 1e-9, cyc_per);
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4);
  run<0, 4>(d, 2000), run<0, 8>(d, 1000), run<1, 4>(d, 2000), run<1, 8>(d, 1000), run<1, 12>(d, 700);
  run<2, 4>(d, 2000), run<2, 8>(d, 1000), run<3, 4>(d, 2000), run<3, 8>(d, 1000), run<3, 12>(d, 700);
  run<4, 8>(d, 1000), run<4, 8>(d, 20000), run<5, 8>(d, 1000), run<5, 8>(d, 5000);
  run<1, 8>(d, 2000), run<6, 8>(d, 2000), run<7, 8>(d, 2000), run<6, 4>(d, 2000), run<7, 4>(d, 2000);
  return 0;
}
