// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths our kernels use
// (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide streaming reads by 2x; other widths are uncalibrated).
// Each kernel streams a known number of bytes once; compare with the counter: scripts/prof_r01.sh calib.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void read_k(const T* __restrict__ p, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    T v = p[i];
    acc += reinterpret_cast<const float*>(&v)[0];
  }
  if (acc == 123.456f) out[0] = acc;
}
template <typename T>
__global__ void write_k(T* p, size_t n) {
  T v{};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int main() {
  const size_t bytes = 1ull << 30;  // 1 GiB, far beyond L2 + Infinity Cache
  void *a, *o;
  hipMalloc(&a, bytes); hipMalloc(&o, 64); hipMemset(a, 0, bytes);
  hipDeviceSynchronize();
  read_k<float><<<2048, 256>>>((const float*)a, bytes / 4, (float*)o);
  read_k<float2><<<2048, 256>>>((const float2*)a, bytes / 8, (float*)o);
  read_k<float4><<<2048, 256>>>((const float4*)a, bytes / 16, (float*)o);
  write_k<float><<<2048, 256>>>((float*)a, bytes / 4);
  write_k<float4><<<2048, 256>>>((float4*)a, bytes / 16);
  hipDeviceSynchronize();
  printf("calib: each kernel moved %zu bytes\n", bytes);
  return 0;
}
