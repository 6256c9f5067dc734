// Wave-level 1024-point complex FFT on packed-fp32 VALU (gfx950, wave64): 16 complex values per lane, Stockham
// radix 16 x 8 x 8 -- three passes, TWO exchanges through a wave-private LDS scratch of 1152 float2.
//
// Every complex value is one 64-bit register pair and every complex add / rotate-by-(-i) / multiply is one or two
// VOP3P instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with op_sel half-swizzles and neg_lo / neg_hi):
//   complex add, sub                         1 instruction
//   a +- (-i) b                              1 instruction (op_sel swaps b's halves, neg flips one of them)
//   complex multiply                         2 instructions
//   radix-4 butterfly 8, radix-8 28 (+14 twiddle multiplies), radix-16 80 instructions
// hipcc builds the packed add / sub from plain vector code; the swizzled forms are inline asm because the compiler
// materialises the swapped operand with v_mov + v_xor instead of folding it into op_sel / neg (checked on ROCm 7.2).
//
// Pass structure for N = 1024 (DIT Stockham; pass with radix R after sub-length NS: butterfly j takes x[j + t N/R],
// multiplies by W_{NS R}^{(j mod NS) t} and writes y[(j / NS) NS R + j mod NS + t NS]):
//   pass 1  R = 16, NS = 1    butterfly = lane; the lane's 16 inputs x[lane + 64 t] ARE its registers; no twiddles.
//                             y[16 lane + m] -> scratch[18 lane + p(m)], p(m) = (m >> 2) + 4 (m & 3) (the register order
//                             the 4 x 4 radix-16 leaves; rows padded 16 -> 18 float2 so the b128 stores are conflict-free)
//   pass 2  R = 8,  NS = 16   butterflies j = lane + 64 u; twiddle row depends on lane & 15 only;
//                             z[i] -> scratch[i + 16 (i >> 7)]
//   pass 3  R = 8,  NS = 128  butterflies ja = lane and jb = 128 - lane (lane 0: 64): X[k] and X[1024 - k] end up in the
//                             SAME lane, so the two-real-signals split needs no cross-lane traffic at all.
#pragma once
#include <hip/hip_runtime.h>

namespace mstpk {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2f mk(float x, float y) {
  v2f r;
  r.x = x, r.y = y;
  return r;
}
// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2f add_mi(v2f a, v2f b) {
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// a - (-i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2f sub_mi(v2f a, v2f b) {
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// complex multiply a * w
__device__ __forceinline__ v2f cmul(v2f a, v2f w) {
  v2f t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                         // (a.x w.x, a.x w.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"                   // (-a.y w.y, a.y w.x) + t
      : "=v"(d) : "v"(a), "v"(w), "v"(t));
  return d;
}
// a * W8^1 = ((a.x + a.y) h, (a.y - a.x) h)
__device__ __forceinline__ v2f mul_w8(v2f a) {
  constexpr float h = 0.70710678118654752440f;
  return add_mi(a, a) * mk(h, h);
}

// radix-4 butterfly, natural order in and out.  ROT2 / ROT3: input a2 / a3 still has to be multiplied by -i (a twiddle
// of the enclosing radix-16 folded into the first additions).
template <bool ROT2 = false, bool ROT3 = false>
__device__ __forceinline__ void dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
  const v2f t0 = ROT2 ? add_mi(a0, a2) : a0 + a2;
  const v2f t1 = ROT2 ? sub_mi(a0, a2) : a0 - a2;
  const v2f t2 = ROT3 ? add_mi(a1, a3) : a1 + a3;
  const v2f d = ROT3 ? sub_mi(a1, a3) : a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = add_mi(t1, d);
  a3 = sub_mi(t1, d);
}

// radix-8 butterfly on v[0..7], natural order in and out
__device__ __forceinline__ void dft8(v2f* v) {
  dft4(v[0], v[2], v[4], v[6]);   // even part -> e0..e3 in v0, v2, v4, v6
  dft4(v[1], v[3], v[5], v[7]);   // odd part  -> o0..o3 in v1, v3, v5, v7
  const v2f o1 = mul_w8(v[3]);    // * W8^1
  const v2f o3 = mul_w8(v[7]);    // * W8^1; the remaining -i of W8^3 rides in add_mi / sub_mi below
  const v2f e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o2 = v[5];
  v[0] = e0 + o0;
  v[4] = e0 - o0;
  v[1] = e1 + o1;
  v[5] = e1 - o1;
  v[2] = add_mi(e2, o2);          // o2 * W8^2 = -i o2
  v[6] = sub_mi(e2, o2);
  v[3] = add_mi(e3, o3);
  v[7] = sub_mi(e3, o3);
}

// radix-16 butterfly on x[0..15] (input t at x[t]); output y[m] is left at x[(m >> 2) + 4 (m & 3)]
__device__ __forceinline__ void dft16(v2f* x) {
  constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;   // cos, sin of pi/8
#pragma unroll
  for (int a = 0; a < 4; ++a) dft4(x[a], x[a + 4], x[a + 8], x[a + 12]);   // over t_b: u[t_a][t_b'] at x[t_a + 4 t_b']
  // twiddles W16^(t_a t_b'); W16^4 = -i and the -i inside W16^6 = -i W8^1 are folded into the second-stage butterflies
  x[1 + 4] = cmul(x[1 + 4], mk(c1, -s1));     // W16^1
  x[1 + 8] = mul_w8(x[1 + 8]);                // W16^2
  x[1 + 12] = cmul(x[1 + 12], mk(s1, -c1));   // W16^3
  x[2 + 4] = mul_w8(x[2 + 4]);                // W16^2
  x[2 + 12] = mul_w8(x[2 + 12]);              // W16^6 = -i W8^1   (-i folded: ROT2 of row 3)
  x[3 + 4] = cmul(x[3 + 4], mk(s1, -c1));     // W16^3
  x[3 + 8] = mul_w8(x[3 + 8]);                // W16^6             (-i folded: ROT3 of row 2)
  x[3 + 12] = cmul(x[3 + 12], mk(-c1, s1));   // W16^9 = -W16^1
  dft4(x[0], x[1], x[2], x[3]);
  dft4(x[4], x[5], x[6], x[7]);
  dft4<true, true>(x[8], x[9], x[10], x[11]);     // x[10]: W16^4 = -i pending; x[11]: -i of W16^6 pending
  dft4<true, false>(x[12], x[13], x[14], x[15]);  // x[14]: -i of W16^6 pending
}

constexpr int kScr = 1152;        // float2 per wave scratch
constexpr int kTw2Rows = 7;       // pass-2 twiddle rows of 64
constexpr int kTw3Rows = 14;      // pass-3 twiddle rows of 64: [u][t-1][lane]

__device__ __forceinline__ int jb_of(int lane) { return lane ? 128 - lane : 64; }

// In: x[t] = element lane + 64 t.  Out: x[t] = X[lane + 128 t], x[8 + t] = X[jb + 128 t], jb = jb_of(lane).
// fill1 / fill2: independent work of the caller, placed right after the reads of exchange 1 / 2 have been issued --
// it executes while the wave would otherwise sit in the LDS round trip.
struct NoFill {
  __device__ __forceinline__ void operator()() const {}
};
template <class F1 = NoFill, class F2 = NoFill>
__device__ __forceinline__ void fft1024(v2f (&x)[16], v2f* scr, const v2f* tw2, const v2f* tw3, int lane, F1 fill1 = F1(),
                                        F2 fill2 = F2()) {
  dft16(x);
  {
    v4f* row = reinterpret_cast<v4f*>(scr + 18 * lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      v4f w;
      w.x = x[2 * q].x, w.y = x[2 * q].y, w.z = x[2 * q + 1].x, w.w = x[2 * q + 1].y;
      row[q] = w;
    }
  }
  __builtin_amdgcn_wave_barrier();
  {
    const int m = lane & 15;
    const v2f* src = scr + 18 * (lane >> 4) + (m >> 2) + 4 * (m & 3);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < 8; ++t) x[8 * u + t] = src[72 * u + 144 * t];
    v2f w2[8];
#pragma unroll
    for (int t = 1; t < 8; ++t) w2[t] = tw2[(t - 1) * 64 + lane];
    fill1();
#pragma unroll
    for (int t = 1; t < 8; ++t) {
      x[t] = cmul(x[t], w2[t]);
      x[8 + t] = cmul(x[8 + t], w2[t]);
    }
    dft8(x);
    dft8(x + 8);
  }
  __builtin_amdgcn_wave_barrier();
  {
    v2f* dst = scr + 144 * (lane >> 4) + (lane & 15);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < 8; ++t) dst[576 * u + 16 * t] = x[8 * u + t];
  }
  __builtin_amdgcn_wave_barrier();
  {
    const v2f* sa = scr + lane;
    const v2f* sb = scr + jb_of(lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = sa[144 * t], x[8 + t] = sb[144 * t];
    fill2();
#pragma unroll
    for (int t = 1; t < 8; ++t) {
      x[t] = cmul(x[t], tw3[(t - 1) * 64 + lane]);
      x[8 + t] = cmul(x[8 + t], tw3[(7 + t - 1) * 64 + lane]);
    }
    dft8(x);
    dft8(x + 8);
  }
}

// host: twiddle tables for fft1024.  tw2[(t-1)*64 + lane] = W_128^((lane & 15) t);
// tw3[(u*7 + t-1)*64 + lane] = W_1024^(j t), j = lane (u = 0) or jb_of(lane) (u = 1).
inline void fill_twiddles_host(float2* tw2, float2* tw3) {
  const double pi = 3.14159265358979323846;
  for (int t = 1; t < 8; ++t)
    for (int lane = 0; lane < 64; ++lane) {
      const double a = -2.0 * pi * (double)((lane & 15) * t) / 128.0;
      tw2[(t - 1) * 64 + lane] = make_float2((float)cos(a), (float)sin(a));
      for (int u = 0; u < 2; ++u) {
        const int j = u ? (lane ? 128 - lane : 64) : lane;
        const double b = -2.0 * pi * (double)(j * t) / 1024.0;
        tw3[(u * 7 + t - 1) * 64 + lane] = make_float2((float)cos(b), (float)sin(b));
      }
    }
}

}  // namespace mstpk
