// Augmentation chain on the device.  Replaces (reference barry-mir/mixing-style-transfer)
// AudioAugmenter.augment_stems src/mixing_utils.py:376-419 and apply_spectral_tilt :421-433, apply_compression
// :435-447, apply_bandwidth_limit :449-456, apply_reverb :458-479.  All random decisions (coins, gain, cutoff,
// impulse response) are drawn by the host in the reference's order and arrive in `mst_aug_clip` / `reverb_ir`.
//
//   aug_chain_kernel   per (clip, channel): x*gain -> biquad (tilt) -> compressor -> 2 biquads (low-pass).
//                      scipy.signal.sosfilt is a sequential fp64 DF2T recurrence; here the channel is cut into
//                      512-sample chunks and solved as a block-parallel state-space scan in fp64:
//                      (1) zero-state response of every chunk in parallel, (2) serial carry of the chunk-boundary
//                      states with A^512, (3) every chunk re-run from its true initial state.  Rounded to fp32
//                      exactly where the reference calls `.float()`.
//   aug_energy_kernel  per-stem mean square for the reverb redistribution weights (:410-416).
//   reverb             y[n] = sum_k ir[k] * mix[n + k - L/2]  (F.conv1d = cross-correlation, :468-474) as a uniformly
//                      partitioned overlap-save convolution: 1024-point wave-level FFTs of 512-sample blocks of
//                      z = mixL + i*mixR (the IR is real, so both channels ride in one complex FFT), 44 spectral
//                      partitions of the reversed IR, spectral multiply-accumulate, inverse FFT, then
//                      rev = 0.7*mix + 0.3*y and stem += rev * (E_stem / (sum E + 1e-8)) * 0.3.
#include "common.h"
#include "fft_wave.h"

namespace {

using namespace mstfft;

constexpr int kLc = 512;      // IIR chunk length
constexpr int kBlk = 512;     // overlap-save block (FFT size 1024)
constexpr int kNfft = 1024;

struct ChainParams {
  float* stems;              // [B][8][T]
  const mst_aug_clip* dec;   // device copy [B]
  double* states;            // [B][8][nchunk][4]
  int T, nchunk;
};

__device__ __forceinline__ float compress_f32(float x) {  // mixing_utils.py:435-447 (threshold -20 dB, ratio 4)
  float db = 20.0f * log10f(fabsf(x) + 1e-8f);
  if (db > -20.0f) db = -20.0f + (db + 20.0f) / 4.0f;
  const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
  return sgn * powf(10.0f, db / 20.0f);
}

// one DF2T biquad step, scipy.signal._sosfilt order; c = {b0,b1,b2,a0,a1,a2}
__device__ __forceinline__ double biquad(const double* c, double* s, double x) {
  const double y = c[0] * x + s[0];
  s[0] = c[1] * x - c[4] * y + s[1];
  s[1] = c[2] * x - c[5] * y;
  return y;
}

template <int NS>
__device__ void run_filter(float* x, int T, int nchunk, double* st, const double* sos, float gain, bool compress,
                           double* M /* shared [4*4] */) {
  constexpr int D = 2 * NS;
  const int tid = threadIdx.x, nth = blockDim.x;
  // (1) zero-state final state of every chunk
  for (int c = tid; c < nchunk; c += nth) {
    double s[D];
#pragma unroll
    for (int i = 0; i < D; ++i) s[i] = 0.0;
    const int n0 = c * kLc, n1 = min(T, n0 + kLc);
    for (int n = n0; n < n1; ++n) {
      double v = (double)(x[n] * gain);
#pragma unroll
      for (int k = 0; k < NS; ++k) v = biquad(sos + 6 * k, s + 2 * k, v);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) st[(size_t)c * 4 + i] = s[i];
  }
  // A^kLc, one column per thread: homogeneous response to a unit initial state
  if (tid < D) {
    double s[D];
#pragma unroll
    for (int i = 0; i < D; ++i) s[i] = (i == tid) ? 1.0 : 0.0;
    for (int n = 0; n < kLc; ++n) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < NS; ++k) v = biquad(sos + 6 * k, s + 2 * k, v);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) M[i * 4 + tid] = s[i];
  }
  __syncthreads();
  // (2) serial carry: st[c] <- state at the START of chunk c
  if (tid == 0) {
    double carry[D];
#pragma unroll
    for (int i = 0; i < D; ++i) carry[i] = 0.0;
    for (int c = 0; c < nchunk; ++c) {
      double zs[D], nx[D];
#pragma unroll
      for (int i = 0; i < D; ++i) zs[i] = st[(size_t)c * 4 + i], st[(size_t)c * 4 + i] = carry[i];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double a = zs[i];
#pragma unroll
        for (int j = 0; j < D; ++j) a += M[i * 4 + j] * carry[j];
        nx[i] = a;
      }
#pragma unroll
      for (int i = 0; i < D; ++i) carry[i] = nx[i];
    }
  }
  __syncthreads();
  // (3) true response, rounded to fp32 where the reference does `.float()`
  for (int c = tid; c < nchunk; c += nth) {
    double s[D];
#pragma unroll
    for (int i = 0; i < D; ++i) s[i] = st[(size_t)c * 4 + i];
    const int n0 = c * kLc, n1 = min(T, n0 + kLc);
    for (int n = n0; n < n1; ++n) {
      double v = (double)(x[n] * gain);
#pragma unroll
      for (int k = 0; k < NS; ++k) v = biquad(sos + 6 * k, s + 2 * k, v);
      float y = (float)v;
      if (compress) y = compress_f32(y);
      x[n] = y;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void aug_chain_kernel(const ChainParams p) {
  __shared__ double M[16];
  __shared__ double sos[18];
  const int b = blockIdx.x >> 3, ch = blockIdx.x & 7;
  const mst_aug_stem& d = p.dec[b].stem[ch >> 1];
  const bool has_g = d.gain != 1.0f, has_t = d.tilt != 0, has_c = d.compress != 0, has_b = d.bw_sections > 0;
  if (!(has_g || has_t || has_c || has_b)) return;
  float* x = p.stems + ((size_t)b * 8 + ch) * p.T;
  double* st = p.states + ((size_t)b * 8 + ch) * p.nchunk * 4;
  if (threadIdx.x < 6) sos[threadIdx.x] = d.tilt_sos[threadIdx.x];
  if (threadIdx.x < 12) sos[6 + threadIdx.x] = d.bw_sos[threadIdx.x];
  __syncthreads();
  if (has_t) {
    run_filter<1>(x, p.T, p.nchunk, st, sos, d.gain, has_c, M);
  } else if (has_g || has_c) {
    for (int n = threadIdx.x; n < p.T; n += blockDim.x) {
      float y = x[n] * d.gain;
      if (has_c) y = compress_f32(y);
      x[n] = y;
    }
    __syncthreads();
  }
  if (has_b) {
    if (d.bw_sections == 1) run_filter<1>(x, p.T, p.nchunk, st, sos + 6, 1.0f, false, M);
    else run_filter<2>(x, p.T, p.nchunk, st, sos + 6, 1.0f, false, M);
  }
}

// mean(stem^2) over (2, T) per stem  ->  redistribution weights E_s / (sum_s E_s + 1e-8)   (:410-416)
constexpr int kEnergyBlocks = 32;  // partial sums per (clip, stem)

__global__ __launch_bounds__(256) void aug_energy_partial_kernel(const float* stems, const mst_aug_clip* dec,
                                                                 double* partial, int T) {
  __shared__ double red[4];
  const int b = blockIdx.y, s = blockIdx.x / kEnergyBlocks, blk = blockIdx.x % kEnergyBlocks;
  if (dec[b].reverb != 1) return;
  const int tid = threadIdx.x;
  const float* x = stems + ((size_t)b * 8 + 2 * s) * T;
  const int n_all = 2 * T, per = (n_all + kEnergyBlocks - 1) / kEnergyBlocks;
  const int n0 = blk * per, n1 = min(n_all, n0 + per);
  double a = 0.0;
  for (int n = n0 + tid; n < n1; n += 256) a += (double)x[n] * (double)x[n];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = a;
  __syncthreads();
  if (tid == 0) partial[((size_t)b * 4 + s) * kEnergyBlocks + blk] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void aug_energy_final_kernel(const mst_aug_clip* dec, const double* partial, float* prop, int T) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0 || dec[b].reverb != 1) return;
  float E[4], tot = 0.f;
  for (int s = 0; s < 4; ++s) {
    double a = 0.0;
    for (int k = 0; k < kEnergyBlocks; ++k) a += partial[((size_t)b * 4 + s) * kEnergyBlocks + k];
    E[s] = (float)(a / (2.0 * T));
    tot += E[s];  // python sum([...]) : (((0 + E0) + E1) + E2) + E3
  }
  tot += 1e-8f;
  for (int s = 0; s < 4; ++s) prop[b * 4 + s] = E[s] / tot;
}

struct RevParams {
  float* stems;               // [B][8][T]
  const mst_aug_clip* dec;
  const float* ir;            // [B][L]
  const float* prop;          // [B][4]
  float2* G;                  // [B][NP][16][64]  spectra of the reversed-IR partitions (register order)
  float2* X;                  // [B][NX][16][64]  spectra of the input windows
  int T, L, NP, NX, D, j0, nj;
};

struct FftLds {
  float2 tw[FftPlan<kNfft>::TW];
  float2 scr[4][kNfft + kNfft / 8];
};

__device__ __forceinline__ float mix_at(const float* stems, size_t clip_off, int T, int c, int n, int mode) {
  const float* v = stems + clip_off + (size_t)c * T;
  if (mode == 2) return v[n];  // plain apply_reverb on the first stem
  return ((v[n] + v[(size_t)2 * T + n]) + v[(size_t)4 * T + n]) + v[(size_t)6 * T + n];  // ((v+b)+d)+o
}

// kind 0: spectra of the reversed IR partitions; kind 1: spectra of the input windows
template <int KIND>
__global__ __launch_bounds__(256) void rev_fft_kernel(const RevParams p) {
  __shared__ FftLds lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  fill_twiddles<kNfft>(lds.tw, tid, 256);
  __syncthreads();
  const int b = blockIdx.y, item = blockIdx.x * 4 + wave;
  const int mode = p.dec[b].reverb;
  if (mode == 0) return;
  const int nitems = KIND == 0 ? p.NP : p.NX;
  if (item >= nitems) return;
  float2 vv[1][16];
  float2 (&v)[16] = vv[0];
  if (KIND == 0) {
    const float* h = p.ir + (size_t)b * p.L;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = lane + 64 * in_q<kNfft>(r);  // time index inside the 1024 window
      const int m = item * kBlk + t;             // index into the reversed IR g[m] = h[L-1-m], first half only
      v[r] = make_float2((t < kBlk && m < p.L) ? h[p.L - 1 - m] : 0.f, 0.f);
    }
  } else {
    const size_t co = (size_t)b * 8 * p.T;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = lane + 64 * in_q<kNfft>(r);
      const int n = kBlk * (item - 1) + t;
      const bool ok = n >= 0 && n < p.T;
      v[r] = ok ? make_float2(mix_at(p.stems, co, p.T, 0, n, mode), mix_at(p.stems, co, p.T, 1, n, mode))
                : make_float2(0.f, 0.f);
    }
  }
  FftPlan<kNfft>::run<1>(vv, lds.scr[wave], lds.tw, lane);
  float2* dst = (KIND == 0 ? p.G + ((size_t)b * p.NP + item) * 1024 : p.X + ((size_t)b * p.NX + item) * 1024) + lane;
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[r * 64] = v[r];
}

__global__ __launch_bounds__(256) void rev_mac_ifft_kernel(const RevParams p) {
  __shared__ FftLds lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  fill_twiddles<kNfft>(lds.tw, tid, 256);
  __syncthreads();
  const int b = blockIdx.y, jj = blockIdx.x * 4 + wave;
  const int mode = p.dec[b].reverb;
  if (mode == 0 || jj >= p.nj) return;
  const int j = p.j0 + jj;
  float2 acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = make_float2(0.f, 0.f);
  const float2* Gb = p.G + (size_t)b * p.NP * 1024 + lane;
  const float2* Xb = p.X + (size_t)b * p.NX * 1024 + lane;
  for (int q = 0; q < p.NP; ++q) {
    const int i = j - q;
    if (i < 0 || i >= p.NX) continue;
    const float2* g = Gb + (size_t)q * 1024;
    const float2* x = Xb + (size_t)i * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float2 a = x[r * 64], w = g[r * 64];
      acc[r].x = fmaf(a.x, w.x, fmaf(-a.y, w.y, acc[r].x));
      acc[r].y = fmaf(a.x, w.y, fmaf(a.y, w.x, acc[r].y));
    }
  }
  // inverse FFT = conj(FFT(conj(Y))) / N; acc is in OUTPUT register order, the FFT wants INPUT register order
  float2 vv[1][16];
  float2 (&v)[16] = vv[0];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float2 y = acc[out_reg<kNfft>(in_q<kNfft>(r))];
    v[r] = make_float2(y.x, -y.y);
  }
  FftPlan<kNfft>::run<1>(vv, lds.scr[wave], lds.tw, lane);
  const float scale = 1.0f / (float)kNfft;
  const size_t co = (size_t)b * 8 * p.T;
  float pr[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) pr[s] = mode == 1 ? p.prop[b * 4 + s] : 0.f;
#pragma unroll
  for (int q = 8; q < 16; ++q) {  // valid overlap-save outputs: window positions 512..1023
    const int t = lane + 64 * q;
    const int n = kBlk * j + (t - kBlk) - p.D;
    if (n < 0 || n >= p.T) continue;
    const float2 z = v[out_reg<kNfft>(q)];
    const float yL = z.x * scale, yR = -z.y * scale;
    const float revL = mix_at(p.stems, co, p.T, 0, n, mode) * 0.7f + yL * 0.3f;  // audio*(1-0.3) + reverb*0.3
    const float revR = mix_at(p.stems, co, p.T, 1, n, mode) * 0.7f + yR * 0.3f;
    if (mode == 2) {
      p.stems[co + n] = revL;
      p.stems[co + (size_t)p.T + n] = revR;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float* xs = p.stems + co + (size_t)(2 * s) * p.T;
        xs[n] = xs[n] + (revL * pr[s]) * 0.3f;
        xs[(size_t)p.T + n] = xs[(size_t)p.T + n] + (revR * pr[s]) * 0.3f;
      }
    }
  }
}

struct AugLayout {
  size_t dec, states, prop, epart, G, X, total;
  int nchunk, NP, NX, D, j0, nj;
};

AugLayout aug_layout(int B, int T, int L) {
  AugLayout a{};
  a.nchunk = (T + kLc - 1) / kLc;
  a.NP = L > 0 ? (L + kBlk - 1) / kBlk : 0;
  a.D = L > 0 ? (L - 1) - L / 2 : 0;
  a.j0 = a.D / kBlk;
  const int j1 = (a.D + T - 1) / kBlk;
  a.nj = L > 0 ? j1 - a.j0 + 1 : 0;
  a.NX = L > 0 ? j1 + 1 : 0;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += mst::align_up(bytes, 256);
    return at;
  };
  a.dec = take((size_t)B * sizeof(mst_aug_clip));
  a.states = take((size_t)B * 8 * a.nchunk * 4 * sizeof(double));
  a.prop = take((size_t)B * 4 * sizeof(float));
  a.epart = take((size_t)B * 4 * kEnergyBlocks * sizeof(double));
  a.G = take((size_t)B * a.NP * 1024 * sizeof(float2));
  a.X = take((size_t)B * a.NX * 1024 * sizeof(float2));
  a.total = o;
  return a;
}

}  // namespace

extern "C" {

size_t mst_aug_workspace_bytes(int B, int T, int ir_len) {
  if (B <= 0 || T <= 0 || ir_len < 0) return 0;
  return aug_layout(B, T, ir_len).total;
}

int mst_aug_apply(const mst_aug_clip* decisions, int B, int T, float* stems_inout, const float* reverb_ir, int ir_len,
                  void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(decisions && stems_inout, "mst_aug_apply: NULL argument");
  MST_REQUIRE(B > 0 && T > 0 && ir_len >= 0, "mst_aug_apply: bad sizes B=%d T=%d ir_len=%d", B, T, ir_len);
  bool any_rev = false;
  for (int b = 0; b < B; ++b) {
    any_rev = any_rev || decisions[b].reverb != 0;
    MST_REQUIRE(decisions[b].reverb >= 0 && decisions[b].reverb <= 2, "mst_aug_apply: bad reverb flag");
    for (int s = 0; s < 4; ++s)
      MST_REQUIRE(decisions[b].stem[s].bw_sections >= 0 && decisions[b].stem[s].bw_sections <= 2,
                  "mst_aug_apply: bw_sections must be 0..2");
  }
  MST_REQUIRE(!any_rev || (reverb_ir && ir_len > 0), "mst_aug_apply: reverb requested but no impulse response");
  const AugLayout L = aug_layout(B, T, any_rev ? ir_len : 0);
  if (!workspace || workspace_bytes < L.total)
    return mst::fail(MST_ENOMEM, "mst_aug_apply: workspace %zu B < required %zu B", workspace_bytes, L.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  mst_aug_clip* ddec = reinterpret_cast<mst_aug_clip*>(ws + L.dec);
  MST_HIP_CHECK(hipMemcpyAsync(ddec, decisions, (size_t)B * sizeof(mst_aug_clip), hipMemcpyHostToDevice, st));
  ChainParams cp{stems_inout, ddec, reinterpret_cast<double*>(ws + L.states), T, L.nchunk};
  hipLaunchKernelGGL(aug_chain_kernel, dim3(B * 8), dim3(256), 0, st, cp);
  MST_HIP_CHECK(hipGetLastError());
  if (any_rev) {
    float* prop = reinterpret_cast<float*>(ws + L.prop);
    double* epart = reinterpret_cast<double*>(ws + L.epart);
    hipLaunchKernelGGL(aug_energy_partial_kernel, dim3(4 * kEnergyBlocks, B), dim3(256), 0, st, stems_inout, ddec, epart, T);
    hipLaunchKernelGGL(aug_energy_final_kernel, dim3(B), dim3(64), 0, st, ddec, epart, prop, T);
    MST_HIP_CHECK(hipGetLastError());
    RevParams rp{stems_inout, ddec, reverb_ir, prop, reinterpret_cast<float2*>(ws + L.G),
                 reinterpret_cast<float2*>(ws + L.X), T, ir_len, L.NP, L.NX, L.D, L.j0, L.nj};
    hipLaunchKernelGGL((rev_fft_kernel<0>), dim3((L.NP + 3) / 4, B), dim3(256), 0, st, rp);
    hipLaunchKernelGGL((rev_fft_kernel<1>), dim3((L.NX + 3) / 4, B), dim3(256), 0, st, rp);
    hipLaunchKernelGGL(rev_mac_ifft_kernel, dim3((L.nj + 3) / 4, B), dim3(256), 0, st, rp);
    MST_HIP_CHECK(hipGetLastError());
  }
  return MST_OK;
}

}  // extern "C"
