// raw buffer loads on gfx950: lanes whose voffset is >= num_records read zeros (no address select, no mask), soffset (scalar)
// moves the whole wave's window and is NOT part of the range check.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_bufload.hip -o scripts/ubench_bufload.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, float* o, unsigned nbytes, unsigned soff) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x % 3 == 1) voff = 0x80000000u;               // "outside": must read zeros
  if (threadIdx.x == 63) voff = nbytes - 8;                    // straddles the end: zeros too
  f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  for (int e = 0; e < 4; ++e) o[threadIdx.x * 4 + e] = f[e];
}
int main() {
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
  float *d, *o;
  (void)hipMalloc(&d, n * 4), (void)hipMalloc(&o, 256 * 4);
  (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  const unsigned nbytes = 2048 * 4, soff = 1024;   // window: 2048 floats; soffset 256 floats
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, nbytes, soff);
  std::vector<float> r(256);
  (void)hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t)
    for (int e = 0; e < 4; ++e) {
      float want = (float)(256 + t * 4 + e + 1);
      if (t % 3 == 1 || t == 63) want = 0.f;
      if (r[t * 4 + e] != want) { if (bad < 8) printf("lane %d elem %d: got %g want %g\n", t, e, r[t * 4 + e], want); ++bad; }
    }
  printf("raw buffer load: %d mismatches (out-of-range lanes read zeros, soffset outside the range check)\n", bad);
  return bad != 0;
}
