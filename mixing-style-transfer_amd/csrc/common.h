// Shared host/device helpers for libmst.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mst.h"

namespace mst {

// thread-local last-error string behind mst_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

#define MST_HIP_CHECK(expr)                                                                   \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return ::mst::fail(MST_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                         __FILE__, __LINE__);                                                 \
  } while (0)

#define MST_REQUIRE(cond, ...)                                                                \
  do {                                                                                        \
    if (!(cond)) return ::mst::fail(MST_EINVAL, __VA_ARGS__);                                 \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// true the first time a call site runs on the CURRENT device: hipFuncSetAttribute (dynamic-LDS limit) is a per-device
// setting, so a process that drives several GPUs has to repeat it on each of them.  `mask` is the call site's static.
inline bool first_use_on_device(unsigned long long& mask) {
  int d = 0;
  (void)hipGetDevice(&d);
  const unsigned long long bit = 1ull << (d & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

template <typename T>
int upload(T** dst, const T* host, size_t n) {
  *dst = nullptr;
  if (n == 0) return MST_OK;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), n * sizeof(T));
  if (e != hipSuccess) return fail(MST_ENOMEM, "hipMalloc(%zu B) failed: %s", n * sizeof(T), hipGetErrorString(e));
  e = hipMemcpy(*dst, host, n * sizeof(T), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(MST_EHIP, "hipMemcpy H2D failed: %s", hipGetErrorString(e));
  return MST_OK;
}

// Blocks b and b+8 share an XCD (round-robin dispatch): give every XCD a contiguous run of
// logical work items so neighbouring tiles hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Order-independent ("deterministic") accumulation for cross-workgroup reductions.  Floating-point atomics make a sum
// depend on the order in which workgroups happen to arrive; integer atomics do not.  A value x is split exactly into
// its integer part and its fraction scaled by 2^48, and both go into 64-bit integer accumulators: the total is the
// same bit pattern whatever the arrival order (the reference asks for deterministic training, src/train.py:30).
// Range of a total |x| < 2^50; resolution 2^-48 absolute (3.6e-15) -- far below fp32 / the double partial sums it replaces;
// up to 2^14 contributions per accumulator without overflow of the fraction word.
// NON-FINITE contributions (a NaN / Inf that entered the trunk, an overflowed AMP loss scale) POISON the accumulator: its integer
// word is raised to kDetPoison and det_get returns NaN, so the statistics, the gradients and everything behind them come out
// non-finite and the caller's NaN checks / GradScaler see the step for what it is (an integer cast of NaN would be garbage that
// looks finite).  The marker survives the integer SUM all-reduce of cross-rank statistics for up to 64 poisoned ranks.
struct DetAcc {
  long long hi, lo;
};
constexpr long long kDetPoison = (1LL << 62) + (1LL << 55), kDetLimit = 1LL << 50;
__device__ __forceinline__ void det_add(DetAcc* a, double x) {
  if (!(fabs(x) < 1.0e15)) {   // NaN, Inf, or beyond the accumulator's range
    atomicMax(&a->hi, kDetPoison);
    return;
  }
  const double hi = rint(x);
  const double lo = rint((x - hi) * 281474976710656.0);   // 2^48; |x - hi| <= 0.5
  atomicAdd(reinterpret_cast<unsigned long long*>(&a->hi), (unsigned long long)(long long)hi);
  atomicAdd(reinterpret_cast<unsigned long long*>(&a->lo), (unsigned long long)(long long)lo);
}
__device__ __forceinline__ double det_get(const DetAcc& a) {
  if (a.hi >= kDetLimit || a.hi <= -kDetLimit) return __builtin_nan("");
  return (double)a.hi + (double)a.lo * (1.0 / 281474976710656.0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Philox-2x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): counter-based, so every element's draw is a
// pure function of (seed, element index) -- no generator state, no order dependence, reproducible for a given seed.
__device__ __forceinline__ uint2 philox2x32(unsigned c0, unsigned c1, unsigned key) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi = __umulhi(0xD256D193u, c0), lo = 0xD256D193u * c0;
    c0 = hi ^ key ^ c1;
    c1 = lo;
    key += 0x9E3779B9u;
  }
  return make_uint2(c0, c1);
}
// Dropout keep decision of element o: kept iff its 32-bit draw >= thresh (= p * 2^32).  (The pair partner o ^ 1 shares the block.)
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, size_t o, unsigned thresh) {
  const uint2 r = philox2x32((unsigned)(o >> 1), (unsigned)(o >> 33) ^ (unsigned)(seed >> 32), (unsigned)seed);
  return ((o & 1) ? r.y : r.x) >= thresh;
}

}  // namespace mst
