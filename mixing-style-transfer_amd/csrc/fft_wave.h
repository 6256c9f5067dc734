// Wave-level complex FFT (gfx950, wave64): a Stockham autosort FFT of NC = 256/512/1024 points held in the
// registers of ONE wave (NC/64 complex values per lane), radix-8/4 butterflies, exchanged between passes through
// a wave-private padded LDS scratch of NC + NC/8 float2.  Shared by stage A (STFT) and the reverb (overlap-save).
#pragma once
#include <hip/hip_runtime.h>

namespace mstfft {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

template <int R>
struct Dft;
template <>
struct Dft<2> {
  static __device__ __forceinline__ void run(float2* v) {
    float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  }
};
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
  float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
  float2 t3 = make_float2(d.y, -d.x);  // (a1 - a3) * (-i)
  a0 = cadd(t0, t2);
  a2 = csub(t0, t2);
  a1 = cadd(t1, t3);
  a3 = csub(t1, t3);
}
template <>
struct Dft<4> {
  static __device__ __forceinline__ void run(float2* v) { dft4(v[0], v[1], v[2], v[3]); }
};
template <>
struct Dft<8> {
  static __device__ __forceinline__ void run(float2* v) {
    constexpr float h = 0.70710678118654752440f;
    dft4(v[0], v[2], v[4], v[6]);  // even part  -> e0..e3 in v0,v2,v4,v6
    dft4(v[1], v[3], v[5], v[7]);  // odd part   -> o0..o3 in v1,v3,v5,v7
    float2 o0 = v[1];
    float2 o1 = make_float2((v[3].x + v[3].y) * h, (v[3].y - v[3].x) * h);   // * W8^1
    float2 o2 = make_float2(v[5].y, -v[5].x);                                // * W8^2 = -i
    float2 o3 = make_float2((v[7].y - v[7].x) * h, -(v[7].x + v[7].y) * h);  // * W8^3
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    v[0] = cadd(e0, o0);
    v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1);
    v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2);
    v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3);
    v[7] = csub(e3, o3);
  }
};

__device__ __forceinline__ int pad8(int e) { return e + (e >> 3); }

// One Stockham pass of radix R with sub-transform length NS (product of earlier radices), applied to NF
// independent FFTs at once (independent dependency chains for the scheduler; twiddles are loaded once).
// Butterfly j = lane + 64*u takes x[j + t*NC/R], multiplies by W_{NS*R}^{(j % NS) t} and writes
// y[(j / NS) * NS * R + j % NS + t * NS].  The LAST pass leaves natural order in registers:
// v[f][u*R + t] = X_f[lane + 64*(u + t*NBF)].  FFT f uses the scratch at scr + f * (NC + NC/8).
template <int NC, int R, int NS, bool FIRST, bool LAST, int NF>
__device__ __forceinline__ void fft_pass(float2 (&v)[NF][NC / 64], float2* scr, const float2* tw, int lane) {
  constexpr int NBF = NC / R / 64;
  constexpr int STR = NC / R;
  constexpr int SCR = NC + NC / 8;
  if constexpr (!FIRST) {
#pragma unroll
    for (int u = 0; u < NBF; ++u)
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const int e = pad8(lane + 64 * u + t * STR);
#pragma unroll
        for (int f = 0; f < NF; ++f) v[f][u * R + t] = scr[f * SCR + e];
      }
  }
#pragma unroll
  for (int u = 0; u < NBF; ++u) {
    if constexpr (NS > 1) {
#pragma unroll
      for (int t = 1; t < R; ++t) {
        // (lane + 64 u) % NS == lane % NS when NS divides 64: those passes store one row per t, not per (u, t)
        const float2 w = tw[((NS <= 64 ? 0 : u) * (R - 1) + (t - 1)) * 64 + lane];
#pragma unroll
        for (int f = 0; f < NF; ++f) v[f][u * R + t] = cmul(v[f][u * R + t], w);
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) Dft<R>::run(&v[f][u * R]);
  }
  if constexpr (!LAST) {
#pragma unroll
    for (int u = 0; u < NBF; ++u) {
      const int j = lane + 64 * u;
      const int base = (j / NS) * NS * R + (j % NS);
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const int e = pad8(base + t * NS);
#pragma unroll
        for (int f = 0; f < NF; ++f) scr[f * SCR + e] = v[f][u * R + t];
      }
    }
  }
}

// Radix plan per complex length.  R0 is the first-pass radix (fixes the register order of the
// windowed input), RL the last (fixes the register order of the spectrum).
template <int NC>
struct FftPlan;
// rows of 64 twiddles a pass contributes to the table
constexpr int tw_rows(int NC, int R, int NS) { return NS == 1 ? 0 : (NS <= 64 ? 1 : NC / R / 64) * (R - 1); }

template <>
struct FftPlan<256> {
  static constexpr int R0 = 4, RL = 4;
  static constexpr int TW = 3 * 3 * 64;
  template <int NF>
  static __device__ __forceinline__ void run(float2 (&v)[NF][4], float2* scr, const float2* tw, int lane) {
    fft_pass<256, 4, 1, true, false, NF>(v, scr, tw, lane);
    fft_pass<256, 4, 4, false, false, NF>(v, scr, tw, lane);
    fft_pass<256, 4, 16, false, false, NF>(v, scr, tw + 3 * 64, lane);
    fft_pass<256, 4, 64, false, true, NF>(v, scr, tw + 2 * 3 * 64, lane);
  }
};
template <>
struct FftPlan<512> {
  static constexpr int R0 = 8, RL = 8;
  static constexpr int TW = 2 * 7 * 64;
  template <int NF>
  static __device__ __forceinline__ void run(float2 (&v)[NF][8], float2* scr, const float2* tw, int lane) {
    fft_pass<512, 8, 1, true, false, NF>(v, scr, tw, lane);
    fft_pass<512, 8, 8, false, false, NF>(v, scr, tw, lane);
    fft_pass<512, 8, 64, false, true, NF>(v, scr, tw + 7 * 64, lane);
  }
};
template <>
struct FftPlan<1024> {
  static constexpr int R0 = 8, RL = 4;
  static constexpr int TW = 7 * 64 + 3 * 64 + 4 * 3 * 64;
  template <int NF>
  static __device__ __forceinline__ void run(float2 (&v)[NF][16], float2* scr, const float2* tw, int lane) {
    fft_pass<1024, 8, 1, true, false, NF>(v, scr, tw, lane);
    fft_pass<1024, 8, 8, false, false, NF>(v, scr, tw, lane);
    fft_pass<1024, 4, 64, false, false, NF>(v, scr, tw + 7 * 64, lane);
    fft_pass<1024, 4, 256, false, true, NF>(v, scr, tw + 7 * 64 + 3 * 64, lane);
  }
};

// Twiddle table for FftPlan<NC>: the concatenation, over the passes with NS > 1, of
// tw[u][t-1][lane] = exp(-2 pi i (j % NS) t / (NS R)),  j = lane + 64 u.  Filled by `count` threads in-kernel
// (double-precision sincospi) or by the host (same formula).
template <int NC>
struct PassList;
template <>
struct PassList<256> {
  static constexpr int N = 4;
  static constexpr int R[4] = {4, 4, 4, 4};
  static constexpr int NS[4] = {1, 4, 16, 64};
};
template <>
struct PassList<512> {
  static constexpr int N = 3;
  static constexpr int R[3] = {8, 8, 8};
  static constexpr int NS[3] = {1, 8, 64};
};
template <>
struct PassList<1024> {
  static constexpr int N = 4;
  static constexpr int R[4] = {8, 8, 4, 4};
  static constexpr int NS[4] = {1, 8, 64, 256};
};

template <int NC>
__device__ inline void fill_twiddles(float2* tw, int tid, int nthreads) {
  using PL = PassList<NC>;
  int base = 0;
#pragma unroll
  for (int ps = 0; ps < PL::N; ++ps) {
    const int R = PL::R[ps], NS = PL::NS[ps];
    if (NS == 1) continue;
    const int nbf = NS <= 64 ? 1 : NC / R / 64, cnt = nbf * (R - 1) * 64;
    for (int e = tid; e < cnt; e += nthreads) {
      const int lane = e & 63, t = (e >> 6) % (R - 1) + 1, u = (e >> 6) / (R - 1);
      const int k = (lane + 64 * u) % NS;
      double sn, cs;
      sincospi(-2.0 * (double)(k * t) / (double)(NS * R), &sn, &cs);
      tw[base + e] = make_float2((float)cs, (float)sn);
    }
    base += cnt;
  }
}

// register index (after the LAST pass) that holds natural-order element lane + 64*q
template <int NC>
__host__ __device__ constexpr int out_reg(int q) {
  constexpr int RL = FftPlan<NC>::RL, NBFL = NC / RL / 64;
  return (q % NBFL) * RL + q / NBFL;
}
// q (element lane + 64*q) that the FIRST pass expects in register r
template <int NC>
__host__ __device__ constexpr int in_q(int r) {
  constexpr int R0 = FftPlan<NC>::R0, NBF0 = NC / R0 / 64;
  return (r / R0) + (r % R0) * NBF0;  // r = u*R0 + t  ->  element lane + 64*(u + t*NBF0)
}

}  // namespace mstfft
