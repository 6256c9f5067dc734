// InfoNCE forward on (all-gathered) embeddings.  Replaces InfoNCELoss.forward src/loss.py:31-136:
//   E = normalize(E); S = E E^T / tau; row-max subtraction; pos_i = sum_{j != i, lab_j == lab_i} exp(S_ij),
//   neg_i = sum_{lab_j != lab_i} exp(S_ij); loss_i = -log(pos_i / (pos_i + neg_i + 1e-8)) for anchors with pos_i > 0.
// One workgroup per local anchor row; no host sync per anchor (the reference syncs N times).
// Backward: d(scale * sum_i loss_i)/dE for all N rows, what autograd derives from the lines above:
//   c_ij = dloss_i/dS_ij = e_ij [j != i] / (pos_i + neg_i + 1e-8) - e_ij [j != i, lab_j == lab_i] / pos_i,
//   dL/de_n = (sum_j c_nj e_j [n local] + sum_i c_in e_i) / tau,  dL/dE_n = (g - e_n (e_n . g)) / |E_n|.
// (The row-max subtraction contributes 1e-8 / (pos + neg + 1e-8) to the arg-max column: below fp32 resolution of
// c_ij, dropped.)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void l2norm_kernel(const float* emb, float* inv_norm, int D) {
  __shared__ float red[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  float a = 0.f;
  for (int d = tid; d < D; d += 256) {
    const float v = emb[(size_t)r * D + d];
    a = fmaf(v, v, a);
  }
  a = mst::wave_sum(a);
  if ((tid & 63) == 0) red[tid >> 6] = a;
  __syncthreads();
  if (tid == 0) inv_norm[r] = 1.0f / fmaxf(sqrtf((red[0] + red[1]) + (red[2] + red[3])), 1e-12f);  // F.normalize eps
}

__global__ __launch_bounds__(256) void infonce_rows_kernel(const float* emb, const int64_t* labels, const float* inv_norm,
                                                           float* sim, int N, int D, int row0, float inv_tau, float* rowout) {
  extern __shared__ float srow[];  // [D] normalised anchor
  __shared__ float red[12];
  const int i = row0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float ni = inv_norm[i];
  for (int d = tid; d < D; d += 256) srow[d] = emb[(size_t)i * D + d] * ni;
  __syncthreads();
  float* s = sim + (size_t)blockIdx.x * N;
  float mx = -INFINITY;
  for (int j0 = wave * 4; j0 < N; j0 += 16) {  // one wave = 4 columns at a time: 4 independent coalesced dot products
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const float* ej[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ej[u] = emb + (size_t)min(j0 + u, N - 1) * D;
    for (int d = lane; d < D; d += 64) {
      const float x = srow[d];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = fmaf(x, ej[u][d], a[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u;
      if (j >= N) break;
      const float v = mst::wave_sum(a[u]) * inv_norm[j] * inv_tau;
      if (lane == 0) s[j] = v;
      mx = fmaxf(mx, v);
    }
  }
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int64_t li = labels[i];
  float pos = 0.f, neg = 0.f;
  for (int j = tid; j < N; j += 256) {
    const float e = expf(s[j] - mx);
    const bool same = labels[j] == li;
    if (same && j != i) pos += e;
    if (!same) neg += e;
  }
  pos = mst::wave_sum(pos), neg = mst::wave_sum(neg);
  if (lane == 0) red[4 + wave] = pos, red[8 + wave] = neg;
  __syncthreads();
  if (tid == 0) {
    pos = (red[4] + red[5]) + (red[6] + red[7]);
    neg = (red[8] + red[9]) + (red[10] + red[11]);
    // per-row result; summed in a fixed order by infonce_reduce_kernel (no float atomics: the loss value is bit-identical
    // from run to run, as the reference's deterministic mode asks -- src/train.py:30)
    rowout[2 * blockIdx.x] = pos > 0.f ? -logf(pos / (pos + neg + 1e-8f)) : 0.f;
    rowout[2 * blockIdx.x + 1] = pos > 0.f ? 1.0f : 0.f;
  }
}

// out[0] = sum of the per-anchor losses, out[1] = number of anchors with a positive: fixed-shape tree, fixed order
__global__ __launch_bounds__(256) void infonce_reduce_kernel(const float* rowout, int rows, float* out) {
  __shared__ float sl[256], sc[256];
  const int tid = threadIdx.x;
  float l = 0.f, c = 0.f;
  for (int r = tid; r < rows; r += 256) l += rowout[2 * r], c += rowout[2 * r + 1];
  sl[tid] = l, sc[tid] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sl[tid] += sl[tid + o], sc[tid] += sc[tid + o];
    __syncthreads();
  }
  if (tid == 0) out[0] = sl[0], out[1] = sc[0];
}

// rows kernel of the backward pass: recomputes S_i., then overwrites it with the coefficients c_i.
__global__ __launch_bounds__(256) void infonce_coef_kernel(const float* emb, const int64_t* labels, const float* inv_norm,
                                                           float* coef, int N, int D, int row0, float inv_tau) {
  extern __shared__ float srow[];
  __shared__ float red[12];
  const int i = row0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float ni = inv_norm[i];
  for (int d = tid; d < D; d += 256) srow[d] = emb[(size_t)i * D + d] * ni;
  __syncthreads();
  float* s = coef + (size_t)blockIdx.x * N;
  float mx = -INFINITY;
  for (int j0 = wave * 4; j0 < N; j0 += 16) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const float* ej[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ej[u] = emb + (size_t)min(j0 + u, N - 1) * D;
    for (int d = lane; d < D; d += 64) {
      const float x = srow[d];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = fmaf(x, ej[u][d], a[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u;
      if (j >= N) break;
      const float v = mst::wave_sum(a[u]) * inv_norm[j] * inv_tau;
      if (lane == 0) s[j] = v;
      mx = fmaxf(mx, v);
    }
  }
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int64_t li = labels[i];
  float pos = 0.f, neg = 0.f;
  for (int j = tid; j < N; j += 256) {
    const float e = expf(s[j] - mx);
    const bool same = labels[j] == li;
    if (same && j != i) pos += e;
    if (!same) neg += e;
  }
  pos = mst::wave_sum(pos), neg = mst::wave_sum(neg);
  if (lane == 0) red[4 + wave] = pos, red[8 + wave] = neg;
  __syncthreads();
  pos = (red[4] + red[5]) + (red[6] + red[7]);
  neg = (red[8] + red[9]) + (red[10] + red[11]);
  const float inv_den = 1.0f / (pos + neg + 1e-8f), inv_pos = pos > 0.f ? 1.0f / pos : 0.f;
  for (int j = tid; j < N; j += 256) {
    const float e = expf(s[j] - mx);
    const bool same = labels[j] == li;
    float c = 0.f;
    if (pos > 0.f && j != i) c = e * inv_den - (same ? e * inv_pos : 0.f);
    s[j] = c;
  }
}

// one workgroup per embedding row n: g = (sum_j c_nj e_j [n local] + sum_i c_in e_i) / tau, then through F.normalize
__global__ __launch_bounds__(256) void infonce_grad_kernel(const float* emb, const float* inv_norm, const float* coef,
                                                           const float* scale, float* grad, int N, int D, int row0,
                                                           int rows, float inv_tau) {
  extern __shared__ float sg[];   // [D] gradient w.r.t. the normalised row
  __shared__ float red[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const bool local = n >= row0 && n < row0 + rows;
  const float* crow = coef + (local ? (size_t)(n - row0) * N : 0);
  float dot = 0.f;
  const float nn = inv_norm[n];
  for (int d = tid; d < D; d += 256) {
    float g = 0.f;
    if (local)
      for (int j = 0; j < N; ++j) g = fmaf(crow[j], emb[(size_t)j * D + d] * inv_norm[j], g);
    for (int i = 0; i < rows; ++i)
      g = fmaf(coef[(size_t)i * N + n], emb[(size_t)(row0 + i) * D + d] * inv_norm[row0 + i], g);
    g *= inv_tau;
    sg[d] = g;
    dot = fmaf(g, emb[(size_t)n * D + d] * nn, dot);
  }
  dot = mst::wave_sum(dot);
  if ((tid & 63) == 0) red[tid >> 6] = dot;
  __syncthreads();
  dot = (red[0] + red[1]) + (red[2] + red[3]);
  const float sc = scale ? *scale : 1.0f;
  for (int d = tid; d < D; d += 256) grad[(size_t)n * D + d] = sc * nn * (sg[d] - emb[(size_t)n * D + d] * nn * dot);
}

}  // namespace

extern "C" {

size_t mst_infonce_workspace_bytes(int N, int D) {
  if (N <= 0 || D <= 0) return 0;
  return mst::align_up((size_t)N * sizeof(float), 256) + mst::align_up((size_t)N * N * sizeof(float), 256) +
         mst::align_up((size_t)2 * N * sizeof(float), 256);
}

int mst_infonce_forward(const float* emb, const int64_t* labels, int N, int D, int row0, int rows, float temperature,
                        float* out, void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(emb && labels && out, "mst_infonce_forward: NULL argument");
  MST_REQUIRE(N > 0 && D > 0 && row0 >= 0 && rows > 0 && row0 + rows <= N && temperature > 0.f,
              "mst_infonce_forward: bad sizes N=%d D=%d row0=%d rows=%d", N, D, row0, rows);
  const size_t need = mst_infonce_workspace_bytes(N, D);
  if (!workspace || workspace_bytes < need)
    return mst::fail(MST_ENOMEM, "mst_infonce_forward: workspace %zu B < required %zu B", workspace_bytes, need);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* inv_norm = reinterpret_cast<float*>(workspace);
  float* sim = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + mst::align_up((size_t)N * sizeof(float), 256));
  float* rowout = reinterpret_cast<float*>(reinterpret_cast<char*>(sim) + mst::align_up((size_t)N * N * sizeof(float), 256));
  hipLaunchKernelGGL(l2norm_kernel, dim3(N), dim3(256), 0, st, emb, inv_norm, D);
  hipLaunchKernelGGL(infonce_rows_kernel, dim3(rows), dim3(256), (size_t)D * sizeof(float), st, emb, labels, inv_norm, sim,
                     N, D, row0, 1.0f / temperature, rowout);
  hipLaunchKernelGGL(infonce_reduce_kernel, dim3(1), dim3(256), 0, st, rowout, rows, out);
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

int mst_infonce_backward(const float* emb, const int64_t* labels, int N, int D, int row0, int rows, float temperature,
                         const float* scale, float* grad, void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(emb && labels && grad, "mst_infonce_backward: NULL argument");
  MST_REQUIRE(N > 0 && D > 0 && row0 >= 0 && rows > 0 && row0 + rows <= N && temperature > 0.f,
              "mst_infonce_backward: bad sizes N=%d D=%d row0=%d rows=%d", N, D, row0, rows);
  const size_t need = mst_infonce_workspace_bytes(N, D);
  if (!workspace || workspace_bytes < need)
    return mst::fail(MST_ENOMEM, "mst_infonce_backward: workspace %zu B < required %zu B", workspace_bytes, need);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* inv_norm = reinterpret_cast<float*>(workspace);
  float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + mst::align_up((size_t)N * sizeof(float), 256));
  hipLaunchKernelGGL(l2norm_kernel, dim3(N), dim3(256), 0, st, emb, inv_norm, D);
  hipLaunchKernelGGL(infonce_coef_kernel, dim3(rows), dim3(256), (size_t)D * sizeof(float), st, emb, labels, inv_norm, coef,
                     N, D, row0, 1.0f / temperature);
  hipLaunchKernelGGL(infonce_grad_kernel, dim3(N), dim3(256), (size_t)D * sizeof(float), st, emb, inv_norm, coef, scale,
                     grad, N, D, row0, rows, 1.0f / temperature);
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

}  // extern "C"
