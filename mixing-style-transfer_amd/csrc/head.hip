// Training forward / backward of the two small networks around the convolution trunk:
//   * the attention-pooling head, reference src/model.py:187-211 behind the Dropout of :118 --
//       xd = Dropout(pool_in);  h = tanh(W0 xd_t + b0);  s_t = w2 . h_t + b2;  a = softmax_t(s);  pooled = sum_t a_t xd_t;
//       emb = Dropout(ReLU(Wp pooled + bp))
//   * the FiLM MLP, reference src/model.py:385-464 --
//       h1 = Dropout(ReLU(W1 f + b1));  h2 = ReLU(W3 h1 + b3);  film = Wh h2 + bh
// with the gradients autograd derives from them.  fp32 throughout (products on v_mfma_f32_16x16x4_f32 where a GEMM is large
// enough to matter, fp32 FMA chains elsewhere); every reduction has a fixed order (no atomics): bit-deterministic.
// Dropout masks are never stored: a keep decision is a pure function of (seed, element index) (Philox-2x32-10, common.h) and
// every kernel that touches a dropped tensor re-derives it.
// The weights are the module's live device tensors in state_dict layout -- nothing is swizzled or cached.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "common.h"
#include "mst.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Drop {
  unsigned long long seed;
  unsigned thresh;   // 0: no dropout
  float scale;       // 1 / (1 - p)
};
__device__ __forceinline__ float drop_apply(const Drop& d, size_t o, float v) {
  if (d.thresh == 0) return v;
  return mst::dropout_keep(d.seed, o, d.thresh) ? v * d.scale : 0.f;
}
Drop make_drop(float p, unsigned long long seed) {
  Drop d{seed, 0u, 1.f};
  if (p > 0.f) {
    const double t = (double)p * 4294967296.0;
    d.thresh = (unsigned)(t > 4294967295.0 ? 4294967295.0 : t);
    d.scale = 1.f / (1.f - p);
  }
  return d;
}

// ------------------------------------------------------------------------------------------
// fp32-MFMA GEMM  C[m][n] = sum_k A(m, k) B(k, n)  with operand FUNCTORS (they return 0 outside the matrix) and an epilogue
// functor.  Workgroup = 4 waves = a 64 x 64 tile; wave w owns rows 16 w .. 16 w + 15 and the four 16-column tiles.
// K runs in chunks of 16 through LDS ([k][64 + 16 words]: the 4 k rows of a fragment read fall into 4 disjoint bank groups).
// AM / BN: the operand's memory is contiguous along m / n (else along k) -- picks the thread -> element mapping of the staging
// loads so that a wave reads runs of consecutive addresses.
// ------------------------------------------------------------------------------------------
constexpr int kGP = 80;   // LDS row pitch (words)
constexpr int kKC = 32;   // K chunk
// blockIdx.z = slice * Z + z: batch entry z (functor argument) and K slice [slice * kslice, (slice + 1) * kslice) -- split-K for the
// GEMMs whose M x N gives too few workgroups to fill the chip; the epilogue then receives the slice's partial sum and a second
// kernel adds the slices in a fixed order (deterministic).
template <bool AM, bool BN, class FA, class FB, class FE>
__global__ __launch_bounds__(256) void gemm_kernel(int M, int N, int K, int Z, int kslice, FA fa, FB fb, FE fe) {
  __shared__ float As[kKC * kGP], Bs[kKC * kGP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64, z = blockIdx.z % Z, slice = blockIdx.z / Z;
  const int kbeg = slice * kslice, kend = min(K, kbeg + kslice);
  const int i = lane & 15, kq = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ra[8], rb[8];
  auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int am = AM ? (tid & 63) : (tid >> 5) + 8 * r, ak = AM ? (tid >> 6) + 4 * r : (tid & 31);
      const int bn = BN ? (tid & 63) : (tid >> 5) + 8 * r, bk = BN ? (tid >> 6) + 4 * r : (tid & 31);
      ra[r] = (m0 + am < M && k0 + ak < kend) ? fa(z, m0 + am, k0 + ak) : 0.f;
      rb[r] = (n0 + bn < N && k0 + bk < kend) ? fb(z, k0 + bk, n0 + bn) : 0.f;
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int am = AM ? (tid & 63) : (tid >> 5) + 8 * r, ak = AM ? (tid >> 6) + 4 * r : (tid & 31);
      const int bn = BN ? (tid & 63) : (tid >> 5) + 8 * r, bk = BN ? (tid >> 6) + 4 * r : (tid & 31);
      As[ak * kGP + am] = ra[r];
      Bs[bk * kGP + bn] = rb[r];
    }
  };
  fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += kKC) {
    __syncthreads();   // the previous chunk's fragment reads are done
    stage();
    __syncthreads();
    if (k0 + kKC < kend) fetch(k0 + kKC);   // in flight behind this chunk's MFMAs
#pragma unroll
    for (int s = 0; s < kKC / 4; ++s) {
      const float a = As[(4 * s + kq) * kGP + 16 * wave + i];
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[(4 * s + kq) * kGP + 16 * n + i], acc[n], 0, 0, 0);
    }
  }
  // D: lane (i, kq) holds rows 16 wave + 4 kq + r, column 16 n + i
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 16 * wave + 4 * kq + r, nn = n0 + 16 * n + i;
      if (m < M && nn < N) fe(z, slice, m, nn, acc[n][r]);
    }
}
// sum of the S slices' partial results [S][count] in slice order, handed to the epilogue functor element by element
template <class FE>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, int S, long long count, FE fe) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += part[(size_t)k * count + i];
  fe(i, s);
}
struct PartEpi {   // partial sums of slice s -> part[s][m][n]
  float* part; int M, N;
  __device__ void operator()(int, int slice, int m, int n, float v) const { part[((size_t)slice * M + m) * N + n] = v; }
};
struct FinHid { float* h; const float* b0; int A; __device__ void operator()(long long i, float v) const { h[i] = tanhf(v + b0[i % A]); } };
struct FinStore { float* out; __device__ void operator()(long long i, float v) const { out[i] = v; } };
struct FinMask {   // d input = sum * (ref > 0 ? gscale : 0): the ReLU (and Dropout) in front of the layer
  float* out; const float* ref; float gscale;
  __device__ void operator()(long long i, float v) const { out[i] = ref[i] > 0.f ? v * gscale : 0.f; }
};
struct FinProj {   // r = ReLU(sum + bias[e]) (kept for the backward), emb = Dropout(r)
  float* emb; float* r; const float* bias; int E; Drop d;
  __device__ void operator()(long long i, float v) const {
    float x = v + bias[i % E];
    x = x < 0.f ? 0.f : x;   // ReLU that keeps NaN (fmaxf would turn a diverged value into 0)
    r[i] = x;
    emb[i] = drop_apply(d, (size_t)i, x);
  }
};
struct LinEpi {   // out[m][n] = Dropout(act(v + bias[n]))
  float* out; const float* bias; int N, relu; Drop d;
  __device__ void operator()(int, int, int m, int n, float v) const {
    float x = v + bias[n];
    if (relu) x = x < 0.f ? 0.f : x;   // (keeps NaN)
    const size_t o = (size_t)m * N + n;
    out[o] = drop_apply(d, o, x);
  }
};
struct RowMajA { const float* a; int ld; __device__ float operator()(int, int m, int k) const { return a[(size_t)m * ld + k]; } };   // A(m, k) = X[m][k]
struct RowMajB { const float* w; int ld; __device__ float operator()(int, int k, int n) const { return w[(size_t)k * ld + n]; } };   // B(k, n) = W[k][n]
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// out[n] = sum_b A[b][n]   (bias gradient): block = 64 columns x 4 row quarters, fixed-order combination
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, float* __restrict__ out, int B, int N) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, n = blockIdx.x * 64 + lane;
  const int per = (B + 3) / 4, b0 = q * per, b1 = min(B, b0 + per);
  float s0 = 0.f, s1 = 0.f;
  if (n < N) {
    int b = b0;
    for (; b + 1 < b1; b += 2) s0 += A[(size_t)b * N + n], s1 += A[(size_t)(b + 1) * N + n];
    if (b < b1) s0 += A[(size_t)b * N + n];
  }
  part[q][lane] = s0 + s1;
  __syncthreads();
  if (q == 0 && n < N) out[n] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// ------------------------------------------------------------------------------------------
// attention pooling
// ------------------------------------------------------------------------------------------
// xd = Dropout(x), materialised once per forward pass (the GEMM operand loaders would otherwise spend more time on the keep
// decisions than the matrix cores on the products)
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ xd, long long n, Drop d) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (i + k < n) xd[i + k] = drop_apply(d, (size_t)(i + k), x[i + k]);
}
// block = (clip, channel slice): scores s_t = b2 + w2 . h_t, softmax over frames -> a[b][t] (slice 0 writes it),
// pooled[b][c] = sum_t a_t xd[b][c][t] for the slice's channels
__global__ __launch_bounds__(256) void head_pool_kernel(const float* __restrict__ x, const float* __restrict__ h, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, float* __restrict__ a_out, float* __restrict__ pooled,
                                                        int C, int T, int A) {   // x = xd (already dropped)
  extern __shared__ float sm[];   // [T] scores -> weights
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = wave; t < T; t += 4) {   // a wave per frame: lanes over the hidden units
    const float* hr = h + ((size_t)b * T + t) * A;
    float s = 0.f;
    for (int j = lane; j < A; j += 64) s = fmaf(w2[j], hr[j], s);
    s = mst::wave_sum(s);
    if (lane == 0) sm[t] = s + b2[0];
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = tid; t < T; t += 256) mx = fmaxf(mx, sm[t]);
  mx = mst::wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int t = tid; t < T; t += 256) {
    const float e = expf(sm[t] - mx);
    sm[t] = e;
    sum += e;
  }
  sum = mst::wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
  if (blockIdx.y == 0)
    for (int t = tid; t < T; t += 256) a_out[(size_t)b * T + t] = sm[t] * inv;
  const int c_per_blk = (C + gridDim.y - 1) / gridDim.y;
  const int c0 = blockIdx.y * c_per_blk, c1 = min(C, c0 + c_per_blk);
  const float* xb = x + (size_t)b * C * T;
  const int sub = lane >> 4, l16 = lane & 15;
  for (int c = c0 + wave * 4 + sub; c < c1 + 3; c += 16) {   // 16 lanes per channel
    float v = 0.f;
    if (c < c1)
      for (int t = l16; t < T; t += 16) v = fmaf(xb[(size_t)c * T + t], sm[t] * inv, v);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (l16 == 0 && c < c1) pooled[(size_t)b * C + c] = v;
  }
}
// dz[b][e] = demb[b][e] * keep3 * scale3 * (r[b][e] > 0)
__global__ __launch_bounds__(256) void head_dz_kernel(const float* __restrict__ demb, const float* __restrict__ r, float* __restrict__ dz,
                                                      long long n, Drop d) {
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  if (o < n) dz[o] = r[o] > 0.f ? drop_apply(d, (size_t)o, demb[o]) : 0.f;
}
// da_t = sum_c dpooled[b][c] xd[b][c][t], in kDaSlices channel slices per clip (block = (clip, slice): lanes over frames, waves
// over the slice's channels) -> dapart[b][slice][t]
constexpr int kDaSlices = 8;
__global__ __launch_bounds__(256) void head_da_kernel(const float* __restrict__ x, const float* __restrict__ dpooled, float* __restrict__ dapart,
                                                      int C, int T) {   // x = xd
  extern __shared__ float sm[];   // [4][T] per-wave partial da
  const int b = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cper = (C + kDaSlices - 1) / kDaSlices, c0 = sl * cper, c1 = min(C, c0 + cper);
  const float* xb = x + (size_t)b * C * T;
  const float* dp = dpooled + (size_t)b * C;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    float s = 0.f;
    if (t < T)
      for (int c = c0 + wave; c < c1; c += 4) s = fmaf(dp[c], xb[(size_t)c * T + t], s);
    if (t < T) sm[wave * T + t] = s;
  }
  __syncthreads();
  for (int t = tid; t < T; t += 256) dapart[((size_t)b * kDaSlices + sl) * T + t] = (sm[t] + sm[T + t]) + (sm[2 * T + t] + sm[3 * T + t]);
}
// block per clip: da_t = sum of the slices (fixed order);  ds_t = a_t (da_t - sum_t' a_t' da_t')
__global__ __launch_bounds__(256) void head_ds_kernel(const float* __restrict__ dapart, const float* __restrict__ a, float* __restrict__ ds, int T) {
  extern __shared__ float sm[];   // [T]
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float dot = 0.f;
  for (int t = tid; t < T; t += 256) {
    float da = 0.f;
    for (int k = 0; k < kDaSlices; ++k) da += dapart[((size_t)b * kDaSlices + k) * T + t];
    sm[t] = da;   // (only this thread touches column t)
    dot = fmaf(a[(size_t)b * T + t], da, dot);
  }
  dot = mst::wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = (red[0] + red[1]) + (red[2] + red[3]);
  for (int t = tid; t < T; t += 256) ds[(size_t)b * T + t] = a[(size_t)b * T + t] * (sm[t] - dot);
}
// block = 64 rows m = (clip, frame): du[m][j] = ds[m] w2[j] (1 - h[m][j]^2);  per-block partial column sums of du (-> db0) and of
// ds h (-> dw2), and the block's sum of ds (-> db2); thread = hidden unit j
__global__ __launch_bounds__(256) void head_du_kernel(const float* __restrict__ h, const float* __restrict__ ds, const float* __restrict__ w2,
                                                      float* __restrict__ du, float* __restrict__ part, int Mrows, int A) {
  const int j = threadIdx.x, m0 = blockIdx.x * 64;
  float s_du = 0.f, s_dw = 0.f, s_ds = 0.f;
  if (j < A) {
    const float w = w2[j];
    for (int r = 0; r < 64; ++r) {
      const int m = m0 + r;
      if (m >= Mrows) break;
      const float hv = h[(size_t)m * A + j], g = ds[m];
      const float v = g * w * (1.f - hv * hv);
      du[(size_t)m * A + j] = v;
      s_du += v;
      s_dw = fmaf(g, hv, s_dw);
      s_ds += g;
    }
  }
  float* pb = part + (size_t)blockIdx.x * (2 * A + 1);
  if (j < A) pb[j] = s_du, pb[A + j] = s_dw;
  if (j == 0) pb[2 * A] = s_ds;
}
// second stage: fixed-order sums of the per-block partials
__global__ __launch_bounds__(256) void head_du_finish_kernel(const float* __restrict__ part, int nblk, int A, float* __restrict__ db0,
                                                             float* __restrict__ dw2, float* __restrict__ db2) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j > 2 * A) return;
  float s = 0.f;
  for (int k = 0; k < nblk; ++k) s += part[(size_t)k * (2 * A + 1) + j];
  if (j < A) db0[j] = s;
  else if (j < 2 * A) dw2[j - A] = s;
  else db2[0] = s;
}

// keep[i] = 1 iff element i survives a Dropout(p) keyed by `seed` (what every kernel above re-derives on the fly)
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* keep, long long n, Drop d) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) keep[i] = (d.thresh == 0 || mst::dropout_keep(d.seed, (size_t)i, d.thresh)) ? 1 : 0;
}

// GEMM functors ---------------------------------------------------------------------------
struct XdOp {   // (m = clip * T + frame, c) -> xd[clip][c][frame]
  const float* x;
  int C, T;
  __device__ float at(int m, int c) const {
    const int b = m / T, t = m - b * T;
    return x[((size_t)b * C + c) * T + t];
  }
};
struct ColMajA { const float* a; int ld; __device__ float operator()(int, int m, int k) const { return a[(size_t)k * ld + m]; } };   // A(m, k) = X[k][m]
struct EpiStore { float* out; int ld; __device__ void operator()(int, int, int m, int n, float v) const { out[(size_t)m * ld + n] = v; } };
struct FwdA { XdOp xd; __device__ float operator()(int, int m, int k) const { return xd.at(m, k); } };
struct RowMajT { const float* w; int ld; __device__ float operator()(int, int k, int n) const { return w[(size_t)n * ld + k]; } };   // B(k, n) = W[n][k]
struct DuT { const float* du; int A; __device__ float operator()(int, int j, int m) const { return du[(size_t)m * A + j]; } };       // A(j, m) = du[m][j]
struct XdB { XdOp xd; __device__ float operator()(int, int m, int c) const { return xd.at(m, c); } };                           // B(m, c)
struct W0T { const float* w; int C; __device__ float operator()(int, int c, int j) const { return w[(size_t)j * C + c]; } };       // A(c, j) = W0[j][c]
struct DuB {   // B(j, t) of clip z = du[(z, t)][j]
  const float* du; int T, A;
  __device__ float operator()(int z, int j, int t) const { return du[((size_t)z * T + t) * A + j]; }
};
struct DxEpi {   // dx[z][c][t] = keep2 scale2 (v + a[z][t] dpooled[z][c])
  float* dx; const float* a; const float* dpooled; int C, T; Drop d;
  __device__ void operator()(int z, int, int c, int t, float v) const {
    const size_t o = ((size_t)z * C + c) * T + t;
    dx[o] = drop_apply(d, o, v + a[(size_t)z * T + t] * dpooled[(size_t)z * C + c]);
  }
};

// C = A B with the K range split over S workgroup slices: partial sums to `part` ([S][M][N]), then a fixed-order sum + epilogue
template <bool AM, bool BN, class FA, class FB, class FIN>
void gemm_split(int M, int N, int K, int S, FA fa, FB fb, float* part, FIN fin, hipStream_t st) {
  const int kslice = round_up((K + S - 1) / S, kKC);
  const int S2 = (K + kslice - 1) / kslice;
  PartEpi pe{part, M, N};
  hipLaunchKernelGGL((gemm_kernel<AM, BN, FA, FB, PartEpi>), dim3((N + 63) / 64, (M + 63) / 64, S2), dim3(256), 0, st, M, N, K, 1, kslice, fa, fb, pe);
  const long long cnt = (long long)M * N;
  hipLaunchKernelGGL((splitk_finish_kernel<FIN>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, part, S2, cnt, fin);
}
constexpr int kSplitProj = 4, kSplitDW0 = 8, kSplitFilm = 8, kSplitHid = 4;

struct HeadSave {   // offsets (floats) into the caller's save buffer
  size_t h, a, pooled, r, part, xd, hpart, total;
};
HeadSave head_save(int B, int C, int T, int A, int E, bool drop_in) {
  HeadSave s{};
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
  s.h = take((size_t)B * T * A), s.a = take((size_t)B * T), s.pooled = take((size_t)B * C), s.r = take((size_t)B * E);
  s.part = take((size_t)kSplitProj * B * E);   // (forward scratch: split-K partial sums of the projection)
  s.xd = take(drop_in ? (size_t)B * C * T : 0);   // Dropout(pool_in), when there is a Dropout
  s.hpart = take((size_t)kSplitHid * B * T * A);   // (forward scratch: split-K partial sums of the hidden layer)
  s.total = o;
  return s;
}
struct HeadWork {   // backward scratch
  size_t dz, dpooled, ds, du, part, gpart, dapart, total;
  int nblk;
};
HeadWork head_work(int B, int C, int T, int A, int E) {
  HeadWork s{};
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
  s.nblk = (B * T + 63) / 64;
  s.dz = take((size_t)B * E), s.dpooled = take((size_t)B * C), s.ds = take((size_t)B * T), s.du = take((size_t)B * T * A);
  s.part = take((size_t)s.nblk * (2 * A + 1));
  s.gpart = take(std::max((size_t)kSplitDW0 * A * C, (size_t)kSplitProj * B * C));   // split-K partial sums
  s.dapart = take((size_t)B * kDaSlices * T);
  s.total = o;
  return s;
}
}  // namespace

// head_pool_kernel / head_ds_kernel / head_da_kernel keep T (4 T for head_da_kernel) floats in dynamic LDS: bounded here, in BOTH
// entry points, so that an over-long clip is refused before any kernel of the pass has run (2048 pooled frames = 4 minutes of audio)
constexpr int kHeadMaxFrames = 2048;
static bool head_dims_ok(int C, int T, int A, int E) {
  return C >= 1 && C <= 3072 && T >= 1 && T <= kHeadMaxFrames && A >= 1 && A <= 256 && E >= 1;
}

extern "C" {

int mst_dropout_mask(float p, uint64_t seed, long long n, unsigned char* keep, void* stream) {
  MST_REQUIRE(keep && n >= 0 && p >= 0.f && p < 1.f, "mst_dropout_mask: bad argument");
  if (n == 0) return MST_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), keep, n,
                     make_drop(p, seed));
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

size_t mst_head_save_bytes(const mst_head_dims* d, int B, float drop_in_p) {
  if (!d || B <= 0) return 0;
  return head_save(B, d->channels, d->frames, d->attn_hidden, d->embed_dim, drop_in_p > 0.f).total * sizeof(float);
}
size_t mst_head_backward_workspace_bytes(const mst_head_dims* d, int B) {
  if (!d || B <= 0) return 0;
  return head_work(B, d->channels, d->frames, d->attn_hidden, d->embed_dim).total * sizeof(float);
}

int mst_head_forward_train(const mst_head_dims* dm, const mst_head_weights* w, const float* pool_in, int B, float drop_in_p,
                           uint64_t drop_in_seed, float drop_out_p, uint64_t drop_out_seed, float* emb, void* save,
                           size_t save_bytes, void* stream) {
  MST_REQUIRE(dm && w && pool_in && emb && save && B > 0, "mst_head_forward_train: NULL / bad argument");
  MST_REQUIRE(w->att0_w && w->att0_b && w->att2_w && w->att2_b && w->proj_w && w->proj_b, "mst_head_forward_train: NULL weight");
  const int C = dm->channels, T = dm->frames, A = dm->attn_hidden, E = dm->embed_dim;
  MST_REQUIRE(head_dims_ok(C, T, A, E), "mst_head_forward_train: dims out of range (C=%d T=%d A=%d E=%d; limits C <= 3072, T <= %d, A <= 256)", C, T, A, E, kHeadMaxFrames);
  MST_REQUIRE(drop_in_p >= 0.f && drop_in_p < 1.f && drop_out_p >= 0.f && drop_out_p < 1.f, "mst_head_forward_train: dropout p must be in [0, 1)");
  const HeadSave S = head_save(B, C, T, A, E, drop_in_p > 0.f);
  if (save_bytes < S.total * sizeof(float)) return mst::fail(MST_ENOMEM, "mst_head_forward_train: save buffer %zu B < %zu B", save_bytes, S.total * sizeof(float));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* sv = static_cast<float*>(save);
  const Drop d2 = make_drop(drop_in_p, drop_in_seed), d3 = make_drop(drop_out_p, drop_out_seed);
  const int M = B * T;
  const float* xd = pool_in;
  if (d2.thresh) {
    const long long n = (long long)B * C * T;
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, st, pool_in, sv + S.xd, n, d2);
    xd = sv + S.xd;
  }
  {   // h = tanh(xd W0^T + b0)
    FwdA fa{XdOp{xd, C, T}};
    RowMajT fb{w->att0_w, C};
    // (split-K: 4 x the workgroups -- enough of them per CU to cover each other's load latency; the partial sums are 25 MB)
    gemm_split<true, false>(M, A, C, kSplitHid, fa, fb, sv + S.hpart, FinHid{sv + S.h, w->att0_b, A}, st);
  }
  hipLaunchKernelGGL(head_pool_kernel, dim3(B, (C + 127) / 128), dim3(256), (size_t)T * sizeof(float), st, xd, sv + S.h, w->att2_w,
                     w->att2_b, sv + S.a, sv + S.pooled, C, T, A);
  // emb = Dropout(ReLU(pooled Wp^T + bp))
  gemm_split<false, false>(B, E, C, kSplitProj, RowMajA{sv + S.pooled, C}, RowMajT{w->proj_w, C}, sv + S.part, FinProj{emb, sv + S.r, w->proj_b, E, d3}, st);
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

int mst_head_backward(const mst_head_dims* dm, const mst_head_weights* w, const float* pool_in, int B, float drop_in_p,
                      uint64_t drop_in_seed, float drop_out_p, uint64_t drop_out_seed, const float* demb, const void* save,
                      const mst_head_grads* g, float* dpool_in, void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(dm && w && pool_in && demb && save && g && dpool_in && workspace && B > 0, "mst_head_backward: NULL / bad argument");
  MST_REQUIRE(g->att0_w && g->att0_b && g->att2_w && g->att2_b && g->proj_w && g->proj_b, "mst_head_backward: NULL gradient pointer");
  const int C = dm->channels, T = dm->frames, A = dm->attn_hidden, E = dm->embed_dim;
  MST_REQUIRE(head_dims_ok(C, T, A, E), "mst_head_backward: dims out of range (C=%d T=%d A=%d E=%d; limits C <= 3072, T <= %d, A <= 256)", C, T, A, E, kHeadMaxFrames);
  MST_REQUIRE(drop_in_p >= 0.f && drop_in_p < 1.f && drop_out_p >= 0.f && drop_out_p < 1.f, "mst_head_backward: dropout p must be in [0, 1)");
  const HeadSave S = head_save(B, C, T, A, E, drop_in_p > 0.f);
  const HeadWork Wk = head_work(B, C, T, A, E);
  if (workspace_bytes < Wk.total * sizeof(float)) return mst::fail(MST_ENOMEM, "mst_head_backward: workspace %zu B < %zu B", workspace_bytes, Wk.total * sizeof(float));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float* sv = static_cast<const float*>(save);
  float* ws = static_cast<float*>(workspace);
  const Drop d2 = make_drop(drop_in_p, drop_in_seed), d3 = make_drop(drop_out_p, drop_out_seed);
  const int M = B * T;
  const float* xd = d2.thresh ? sv + S.xd : pool_in;
  // projection
  hipLaunchKernelGGL(head_dz_kernel, dim3((unsigned)(((long long)B * E + 255) / 256)), dim3(256), 0, st, demb, sv + S.r, ws + Wk.dz, (long long)B * E, d3);
  hipLaunchKernelGGL(colsum_kernel, dim3((E + 63) / 64), dim3(256), 0, st, ws + Wk.dz, g->proj_b, B, E);
  hipLaunchKernelGGL((gemm_kernel<true, true, ColMajA, RowMajB, EpiStore>), dim3((C + 63) / 64, (E + 63) / 64, 1), dim3(256), 0, st, E, C, B, 1,
                     round_up(B, kKC), ColMajA{ws + Wk.dz, E}, RowMajB{sv + S.pooled, C}, EpiStore{g->proj_w, C});   // dWp = dz^T pooled
  gemm_split<false, true>(B, C, E, kSplitProj, RowMajA{ws + Wk.dz, E}, RowMajB{w->proj_w, C}, ws + Wk.gpart, FinStore{ws + Wk.dpooled}, st);   // dpooled = dz Wp
  // softmax / scores
  hipLaunchKernelGGL(head_da_kernel, dim3(B, kDaSlices), dim3(256), (size_t)4 * T * sizeof(float), st, xd, ws + Wk.dpooled, ws + Wk.dapart, C, T);
  hipLaunchKernelGGL(head_ds_kernel, dim3(B), dim3(256), (size_t)T * sizeof(float), st, ws + Wk.dapart, sv + S.a, ws + Wk.ds, T);
  hipLaunchKernelGGL(head_du_kernel, dim3(Wk.nblk), dim3(256), 0, st, sv + S.h, ws + Wk.ds, w->att2_w, ws + Wk.du, ws + Wk.part, M, A);
  hipLaunchKernelGGL(head_du_finish_kernel, dim3((2 * A + 1 + 255) / 256), dim3(256), 0, st, ws + Wk.part, Wk.nblk, A, g->att0_b, g->att2_w, g->att2_b);
  {   // dW0[j][c] = sum_m du[m][j] xd[m][c]
    DuT fa{ws + Wk.du, A};
    XdB fb{XdOp{xd, C, T}};
    gemm_split<true, false>(A, C, M, kSplitDW0, fa, fb, ws + Wk.gpart, FinStore{g->att0_w}, st);
  }
  {   // d pool_in[z][c][t] = keep (a[z][t] dpooled[z][c] + sum_j W0[j][c] du[(z, t)][j])
    W0T fa{w->att0_w, C};
    DuB fb{ws + Wk.du, T, A};
    DxEpi fe{dpool_in, sv + S.a, ws + Wk.dpooled, C, T, d2};
    hipLaunchKernelGGL((gemm_kernel<true, false, W0T, DuB, DxEpi>), dim3((T + 63) / 64, (C + 63) / 64, B), dim3(256), 0, st, C, T, A, B,
                       round_up(A, kKC), fa, fb, fe);
  }
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

size_t mst_film_save_bytes(const mst_film_dims* d, int B) {
  if (!d || B <= 0) return 0;
  return (size_t)2 * B * d->hidden * sizeof(float);
}
size_t mst_film_backward_workspace_bytes(const mst_film_dims* d, int B) {
  if (!d || B <= 0) return 0;
  return (size_t)(2 + kSplitFilm) * B * d->hidden * sizeof(float);   // dh2, dh1, split-K partial sums
}

int mst_film_forward_train(const mst_film_dims* dm, const mst_film_weights* w, const float* feats, int B, float drop_p, uint64_t drop_seed,
                           float* film, void* save, size_t save_bytes, void* stream) {
  MST_REQUIRE(dm && w && feats && film && save && B > 0, "mst_film_forward_train: NULL / bad argument");
  MST_REQUIRE(w->mlp0_w && w->mlp0_b && w->mlp3_w && w->mlp3_b && w->head_w && w->head_b, "mst_film_forward_train: NULL weight");
  const int Fd = dm->feature_dim, H = dm->hidden, O = dm->out_dim;
  MST_REQUIRE(Fd >= 1 && Fd <= 2048 && H >= 1 && H <= 2048 && O >= 1, "mst_film_forward_train: dims out of range");
  MST_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mst_film_forward_train: dropout p must be in [0, 1)");
  if (save_bytes < mst_film_save_bytes(dm, B)) return mst::fail(MST_ENOMEM, "mst_film_forward_train: save buffer too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* h1d = static_cast<float*>(save);
  float* h2 = h1d + (size_t)B * H;
  auto linear = [&](const float* X, const float* W, const float* bias, float* out, int N, int K, int relu, Drop d) {   // out = act(X W^T + b)
    hipLaunchKernelGGL((gemm_kernel<false, false, RowMajA, RowMajT, LinEpi>), dim3((N + 63) / 64, (B + 63) / 64, 1), dim3(256), 0, st, B, N, K, 1,
                       round_up(K, kKC), RowMajA{X, K}, RowMajT{W, K}, LinEpi{out, bias, N, relu, d});
  };
  linear(feats, w->mlp0_w, w->mlp0_b, h1d, H, Fd, 1, make_drop(drop_p, drop_seed));
  linear(h1d, w->mlp3_w, w->mlp3_b, h2, H, H, 1, make_drop(0.f, 0));
  linear(h2, w->head_w, w->head_b, film, O, H, 0, make_drop(0.f, 0));
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

int mst_film_backward(const mst_film_dims* dm, const mst_film_weights* w, const float* feats, int B, float drop_p, const float* dfilm,
                      const void* save, const mst_film_grads* g, void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(dm && w && feats && dfilm && save && g && workspace && B > 0, "mst_film_backward: NULL / bad argument");
  MST_REQUIRE(g->mlp0_w && g->mlp0_b && g->mlp3_w && g->mlp3_b && g->head_w && g->head_b, "mst_film_backward: NULL gradient pointer");
  const int Fd = dm->feature_dim, H = dm->hidden, O = dm->out_dim;
  if (workspace_bytes < mst_film_backward_workspace_bytes(dm, B)) return mst::fail(MST_ENOMEM, "mst_film_backward: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float* h1d = static_cast<const float*>(save);
  const float* h2 = h1d + (size_t)B * H;
  float* dh2 = static_cast<float*>(workspace);
  float* dh1 = dh2 + (size_t)B * H;
  float* part = dh1 + (size_t)B * H;
  const dim3 blk(256);
  // film_head
  hipLaunchKernelGGL(colsum_kernel, dim3((O + 63) / 64), blk, 0, st, dfilm, g->head_b, B, O);
  hipLaunchKernelGGL((gemm_kernel<true, true, ColMajA, RowMajB, EpiStore>), dim3((H + 63) / 64, (O + 63) / 64, 1), blk, 0, st, O, H, B, 1,
                     round_up(B, kKC), ColMajA{dfilm, O}, RowMajB{h2, H}, EpiStore{g->head_w, H});   // dWh = dfilm^T h2
  gemm_split<false, true>(B, H, O, kSplitFilm, RowMajA{dfilm, O}, RowMajB{w->head_w, H}, part, FinMask{dh2, h2, 1.f}, st);   // through the ReLU
  // feature_mlp.3
  hipLaunchKernelGGL(colsum_kernel, dim3((H + 63) / 64), blk, 0, st, dh2, g->mlp3_b, B, H);
  hipLaunchKernelGGL((gemm_kernel<true, true, ColMajA, RowMajB, EpiStore>), dim3((H + 63) / 64, (H + 63) / 64, 1), blk, 0, st, H, H, B, 1,
                     round_up(B, kKC), ColMajA{dh2, H}, RowMajB{h1d, H}, EpiStore{g->mlp3_w, H});
  // through Dropout and ReLU: the survivors are exactly the positive entries of the dropped activation
  gemm_split<false, true>(B, H, H, 2, RowMajA{dh2, H}, RowMajB{w->mlp3_w, H}, part, FinMask{dh1, h1d, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f}, st);
  // feature_mlp.0
  hipLaunchKernelGGL(colsum_kernel, dim3((H + 63) / 64), blk, 0, st, dh1, g->mlp0_b, B, H);
  hipLaunchKernelGGL((gemm_kernel<true, true, ColMajA, RowMajB, EpiStore>), dim3((Fd + 63) / 64, (H + 63) / 64, 1), blk, 0, st, H, Fd, B, 1,
                     round_up(B, kKC), ColMajA{dh1, H}, RowMajB{feats, Fd}, EpiStore{g->mlp0_w, Fd});
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

}  // extern "C"
