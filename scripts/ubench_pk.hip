// Micro-benchmark (gfx950): issue cost of packed-fp32 VALU instructions vs scalar fp32, per wave-instruction, at 1..3
// waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -o scripts/ubench_pk.bin scripts/ubench_pk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ void kern(float* out, long long* cyc, int iters) {
  v2f a0 = {1.0f + threadIdx.x, 0.5f}, a1 = {0.25f, 1.5f}, a2 = {0.75f, 0.1f}, a3 = {0.3f, 0.2f};
  v2f a4 = a0 * 0.5f, a5 = a1 * 0.5f, a6 = a2 * 0.5f, a7 = a3 * 0.5f;
  v2f w = {0.999f, 0.01f};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {   // 8 independent v_fma_f32 chains x 8
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                        "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                        : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(w.x));)
    } else if (MODE == 1) {  // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                        "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));)
    } else if (MODE == 2) {  // v_pk_add_f32
      REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                        "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));)
    } else if (MODE == 3) {  // v_pk_mul_f32 with op_sel broadcast
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %1, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %2, %2, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %3, %8 op_sel_hi:[0,1]\n"
                        "v_pk_mul_f32 %4, %4, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %5, %5, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %6, %6, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %7, %7, %8 op_sel_hi:[0,1]\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));)
    } else if (MODE == 4) {  // v_pk_fma_f32 with swizzle + neg (second half of a complex multiply)
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %1, %1, %8, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
                        "v_pk_fma_f32 %2, %2, %8, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %3, %3, %8, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
                        "v_pk_fma_f32 %4, %4, %8, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %5, %5, %8, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
                        "v_pk_fma_f32 %6, %6, %8, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %7, %7, %8, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));)
    } else if (MODE == 5) {  // v_add_f32 scalar
      REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                        "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                        : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(w.x));)
    } else if (MODE == 6) {  // dependent chain of v_pk_fma (latency)
      REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n" : "+v"(a0) : "v"(w));)
    } else if (MODE == 7) {  // dependent chain of v_fma (latency)
      REP64(asm volatile("v_fma_f32 %0, %0, %1, %0\n" : "+v"(a0.x) : "v"(w.x));)
    }
  }
  long long t1 = __builtin_readcyclecounter();
  v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMalloc(&cyc, 8);
  const int iters = 2000, threads = 256 * waves_per_simd;   // one workgroup per CU, `waves_per_simd` waves on each SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<MODE><<<256, threads>>>(out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<MODE><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = 64.0 * iters;   // instructions per wave
  printf("%-34s waves/SIMD %d: %7.2f counter-ticks per wave-instr (wave 0); %6.3f ns per instr per SIMD (wall: %.3f ms)\n", name,
         waves_per_simd, (double)c / n, ms * 1e6 / (n * waves_per_simd), ms);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 3; ++w) {
    run<0>("v_fma_f32 (8 indep chains)", w);
    run<1>("v_pk_fma_f32", w);
    run<2>("v_pk_add_f32", w);
    run<3>("v_pk_mul_f32 op_sel bcast", w);
    run<4>("v_pk_fma_f32 op_sel swz + neg_lo", w);
    run<5>("v_add_f32", w);
  }
  run<6>("v_pk_fma_f32 dependent chain", 1);
  run<7>("v_fma_f32 dependent chain", 1);
  return 0;
}
