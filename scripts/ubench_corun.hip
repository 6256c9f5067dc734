// What a wave OUTSIDE its matrix phase gets when its SIMD partner is INSIDE one (gfx950).  8 waves per workgroup, one workgroup
// per CU: waves 0..3 run conv1_resident_kernel's k-step pattern (fp32 16x16x4 MFMAs with one ds_read_b32 after each of the first
// seven); waves 4..7 run a block of 192 instructions of one kind, timed with s_memtime, over and over until the MFMA waves
// are done.  KIND 0: v_cndmask + ds_write_b32 pairs; 1: VALU only (v_fma chain); 2: SALU only (s_add chain); 3: ds_write_b32 only;
// 4: v_cndmask + ds_write pairs with the MFMA waves IDLE (sleeping) for reference.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_corun.hip -o scripts/ubench_corun.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  __shared__ float smem[8 * 2304];
  __shared__ int done;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8 * 2304; i += 512) smem[i] = (float)(i & 7) * 0.125f;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  float res = 0.f;
  if (wave < 4) {
    if (KIND == 4) {
      for (int it = 0; it < iters * 40; ++it) __builtin_amdgcn_s_sleep(100);
    } else {
      const float* pb = smem + wave * 2304 + lane;
      f32x4 acc[5][2];
      for (int t = 0; t < 5; ++t)
        for (int n = 0; n < 2; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      float a[2][5], b[2][2];
      for (int t = 0; t < 5; ++t) a[0][t] = a[1][t] = pb[64 * t];
      for (int n = 0; n < 2; ++n) b[0][n] = b[1][n] = pb[64 * (5 + n)];
      const long long t0 = clock64();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
          const int cu = ks & 1, nx = cu ^ 1;
#pragma unroll
          for (int i = 0; i < 10; ++i) {
            acc[i / 2][i % 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cu][i / 2], b[cu][i % 2], acc[i / 2][i % 2], 0, 0, 0);
            if (i < 2) b[nx][i] = pb[64 * (5 + i) + ks];
            else if (i < 7) a[nx][i - 2] = pb[64 * (i - 2) + ks];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      const long long t1 = clock64();
      if (lane == 0) cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0, cyc[(blockIdx.x * 8 + wave) * 2 + 1] = (long long)iters * 140;
      for (int t = 0; t < 5; ++t)
        for (int n = 0; n < 2; ++n) res += acc[t][n][0] + acc[t][n][1] + acc[t][n][2] + acc[t][n][3];
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) atomicAdd(&done, 1);
  } else {
    float* pw = smem + wave * 2304 + lane;
    long long total = 0, blocks = 0;
    float v = (float)lane, w = 1.0001f;
    int sv = __builtin_amdgcn_readfirstlane(wave);
    while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
      const long long t0 = clock64();
      if (KIND == 0 || KIND == 4) {
#pragma unroll
        for (int i = 0; i < 96; ++i) {
          float x;
          asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(v), "v"(w));
          asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"((unsigned)(size_t)pw), "v"(x), "n"((i % 32) * 256) : "memory");
        }
      } else if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 192; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(w));
      } else if (KIND == 2) {
#pragma unroll
        for (int i = 0; i < 192; ++i) asm volatile("s_add_i32 %0, %0, 3" : "+s"(sv));
      } else {
#pragma unroll
        for (int i = 0; i < 192; ++i)
          asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"((unsigned)(size_t)pw), "v"(v), "n"((i % 32) * 256) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      total += clock64() - t0, ++blocks;
    }
    res = v + (float)sv;
    if (lane == 0) cyc[(blockIdx.x * 8 + wave) * 2] = total, cyc[(blockIdx.x * 8 + wave) * 2 + 1] = blocks;
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}
template <int KIND>
void run(float* d, long long* c, const char* name) {
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, c, 300);
  (void)hipDeviceSynchronize();
  static long long h[256 * 16];
  (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double mf = 0, mfn = 0, ot = 0, on = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) {
      if (w < 4) mf += h[(b * 8 + w) * 2], mfn += h[(b * 8 + w) * 2 + 1];
      else ot += h[(b * 8 + w) * 2], on += h[(b * 8 + w) * 2 + 1];
    }
  printf("%-44s MFMA waves: %.1f cycles per MFMA   partner: %.0f cycles per block of 192 instructions = %.1f per instruction\n", name,
         KIND == 4 ? 0.0 : mf / mfn, ot / on, ot / on / 192);
}
int main() {
  float* d;
  long long* c;
  (void)hipMalloc(&d, 256 * 512 * 4);
  (void)hipMalloc(&c, 256 * 16 * 8);
  run<4>(d, c, "v_cndmask + ds_write_b32, partner idle");
  run<0>(d, c, "v_cndmask + ds_write_b32");
  run<1>(d, c, "v_fma_f32 chain");
  run<2>(d, c, "s_add_i32 chain");
  run<3>(d, c, "ds_write_b32");
  return 0;
}
