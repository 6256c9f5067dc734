// Augmentation chain on the device.  Replaces (reference barry-mir/mixing-style-transfer)
// AudioAugmenter.augment_stems src/mixing_utils.py:376-419 and apply_spectral_tilt :421-433, apply_compression
// :435-447, apply_bandwidth_limit :449-456, apply_reverb :458-479.  All random decisions (coins, gain, cutoff,
// impulse response) are drawn by the host in the reference's order and arrive in `mst_aug_clip` / `reverb_ir`.
//
//   aug_tilt_zs / aug_scan<0> / aug_tilt_resp / aug_scan<1> / aug_bw_resp: x*gain -> biquad (tilt) -> compressor
//                      -> 1-2 biquads (low-pass), thread per 512-sample chunk, float64 block-parallel IIR.
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

// ---- block-parallel IIR: chunk c of a stream = samples [c*kLc, (c+1)*kLc).
//   zero-state pass (thread per chunk)  ->  z_c = final filter state of the chunk started from rest
//   scan (one wave per stream)          ->  start state of every chunk: s_{c+1} = A^kLc s_c + z_c
//   response pass (thread per chunk)    ->  true output from the start state
// The chain is: x*gain -> [tilt biquad] -> [compressor] -> store -> [1-2 low-pass biquads]; the tilt response pass
// also runs the low-pass zero-state pass on the values it has just produced (one read of the waveform less).
// Every sample goes through exactly the reference's arithmetic (float64 DF2T, scipy sosfilt order, `.float()` after
// each filter, mixing_utils.py:421-456); only the chunk start states are obtained differently.
struct Coef {   // one biquad of an sos row {b0,b1,b2,a0,a1,a2}: b0, b1, b2, a1, a2
  double b0, b1, b2, a1, a2;
};
__device__ __forceinline__ Coef coef_of(const double* c) { return Coef{c[0], c[1], c[2], c[4], c[5]}; }
__device__ __forceinline__ double biquad_step(const Coef& c, double& s0, double& s1, double x) {
  const double y = c.b0 * x + s0;
  s0 = c.b1 * x - c.a1 * y + s1;
  s1 = c.b2 * x - c.a2 * y;
  return y;
}

// walk one chunk 4 samples at a time (16-byte loads, next group prefetched) or sample by sample when the chunk is
// ragged / unaligned; f(value) returns the value to store, STORE selects whether it is written back
template <bool STORE, typename F>
__device__ __forceinline__ void walk_chunk(float* x, int n0, int n1, F&& f) {
  if (n1 - n0 == kLc && (reinterpret_cast<uintptr_t>(x + n0) & 15) == 0) {
    float4 cur = *reinterpret_cast<const float4*>(x + n0);
#pragma unroll 2
    for (int i = 0; i < kLc; i += 4) {
      const float4 nxt = *reinterpret_cast<const float4*>(x + n0 + min(i + 4, kLc - 4));
      float4 o;
      o.x = f(cur.x), o.y = f(cur.y), o.z = f(cur.z), o.w = f(cur.w);
      if (STORE) *reinterpret_cast<float4*>(x + n0 + i) = o;
      cur = nxt;
    }
  } else {
    for (int n = n0; n < n1; ++n) {
      const float o = f(x[n]);
      if (STORE) x[n] = o;
    }
  }
}

// grid (ceil(nchunk/256), B*8): zero-state pass of the tilt biquad on x*gain
__global__ __launch_bounds__(256) void aug_tilt_zs_kernel(const ChainParams p) {
  const int stream = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  const mst_aug_stem& d = p.dec[stream >> 3].stem[(stream & 7) >> 1];
  if (d.tilt == 0 || c >= p.nchunk) return;
  float* x = p.stems + (size_t)stream * p.T;
  const Coef k = coef_of(d.tilt_sos);
  const float gain = d.gain;
  double s0 = 0.0, s1 = 0.0;
  walk_chunk<false>(x, c * kLc, min(p.T, (c + 1) * kLc), [&](float v) {
    (void)biquad_step(k, s0, s1, (double)(v * gain));
    return 0.f;
  });
  double* st = p.states + ((size_t)stream * p.nchunk + c) * 4;
  st[0] = s0, st[1] = s1;
}

// grid (B*8), one wave: chunk start states from the zero-state finals.  WHICH 0: tilt (1 section), 1: low-pass (1-2).
// Two levels: every lane folds its G = ceil(nchunk/64) consecutive chunks, lane 0 chains the 64 groups with M^G, every
// lane then replays its chunks from its group's start state (serial depth 2G + 64 instead of nchunk).
template <int WHICH>
__global__ __launch_bounds__(64) void aug_scan_kernel(const ChainParams p) {
  __shared__ double Msh[16], MGsh[16], agg[64 * 4], cin[64 * 4];
  const int stream = blockIdx.x, lane = threadIdx.x;
  const mst_aug_stem& d = p.dec[stream >> 3].stem[(stream & 7) >> 1];
  const int NS = WHICH == 0 ? (d.tilt != 0 ? 1 : 0) : d.bw_sections;
  if (NS == 0) return;   // block-uniform
  const int D = 2 * NS;
  const double* sos = WHICH == 0 ? d.tilt_sos : d.bw_sos;
  double* st = p.states + (size_t)stream * p.nchunk * 4;
  if (lane < 16) Msh[lane] = 0.0;
  __syncthreads();
  // M = A^kLc, one column per lane: homogeneous response to a unit initial state
  if (lane < D) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < 4; ++i) s[i] = (i == lane) ? 1.0 : 0.0;
    const Coef k0 = coef_of(sos), k1 = coef_of(sos + (NS > 1 ? 6 : 0));
    for (int n = 0; n < kLc; ++n) {
      double v = biquad_step(k0, s[0], s[1], 0.0);
      if (NS > 1) v = biquad_step(k1, s[2], s[3], v);
    }
    for (int i = 0; i < D; ++i) Msh[i * 4 + lane] = s[i];
  }
  __syncthreads();
  double M[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) M[i][j] = Msh[i * 4 + j];   // zero outside the D x D block
  const int G = (p.nchunk + 63) / 64;
  const int c0 = min(p.nchunk, lane * G), c1 = min(p.nchunk, c0 + G);
  auto step = [&](double (&a)[4], int c) {   // a <- M a + z_c
    double nx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double t = i < D ? st[(size_t)c * 4 + i] : 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) t += M[i][j] * a[j];
      nx[i] = t;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = nx[i];
  };
  // (1) fold this lane's chunks from rest; lane 0 also forms M^G
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int c = c0; c < c1; ++c) step(a, c);
#pragma unroll
  for (int i = 0; i < 4; ++i) agg[lane * 4 + i] = a[i];
  if (lane == 0) {
    double P[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) P[i][j] = M[i][j];
    for (int g = 1; g < G; ++g) {
      double Q[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double t = 0.0;
#pragma unroll
          for (int k = 0; k < 4; ++k) t += M[i][k] * P[k][j];
          Q[i][j] = t;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) P[i][j] = Q[i][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) MGsh[i * 4 + j] = P[i][j];
  }
  __syncthreads();
  // (2) chain the 64 groups
  if (lane == 0) {
    double cr[4] = {0.0, 0.0, 0.0, 0.0};
    for (int g = 0; g < 64; ++g) {
      double nx[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        cin[g * 4 + i] = cr[i];
        double t = agg[g * 4 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) t += MGsh[i * 4 + j] * cr[j];
        nx[i] = t;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) cr[i] = nx[i];
    }
  }
  __syncthreads();
  // (3) start state of every chunk of the group (overwrites z_c)
  double sv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) sv[i] = cin[lane * 4 + i];
  for (int c = c0; c < c1; ++c) {
    double keep[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) keep[i] = sv[i];
    step(sv, c);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < D) st[(size_t)c * 4 + i] = keep[i];
  }
}

// grid (ceil(nchunk/256), B*8): gain -> tilt response -> compressor -> store, then the low-pass zero-state pass on
// the stored values
__global__ __launch_bounds__(256) void aug_tilt_resp_kernel(const ChainParams p) {
  const int stream = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  const mst_aug_stem& d = p.dec[stream >> 3].stem[(stream & 7) >> 1];
  const bool has_g = d.gain != 1.0f, has_t = d.tilt != 0, has_c = d.compress != 0, has_b = d.bw_sections > 0;
  if (!(has_g || has_t || has_c || has_b) || c >= p.nchunk) return;
  float* x = p.stems + (size_t)stream * p.T;
  double* st = p.states + ((size_t)stream * p.nchunk + c) * 4;
  const Coef kt = coef_of(d.tilt_sos), kb0 = coef_of(d.bw_sos), kb1 = coef_of(d.bw_sos + 6);
  double t0 = has_t ? st[0] : 0.0, t1 = has_t ? st[1] : 0.0;
  double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
  const float gain = d.gain;
  const bool two = d.bw_sections > 1;
  auto body = [&](float v) {
    float y = v * gain;
    if (has_t) y = (float)biquad_step(kt, t0, t1, (double)y);
    if (has_c) y = compress_f32(y);
    if (has_b) {
      const double w = biquad_step(kb0, b0, b1, (double)y);
      if (two) (void)biquad_step(kb1, b2, b3, w);
    }
    return y;
  };
  const int n0 = c * kLc, n1 = min(p.T, n0 + kLc);
  if (has_g || has_t || has_c) walk_chunk<true>(x, n0, n1, body);
  else walk_chunk<false>(x, n0, n1, body);
  if (has_b) st[0] = b0, st[1] = b1, st[2] = b2, st[3] = b3;
}

// grid (ceil(nchunk/256), B*8): low-pass response from the scanned start states
__global__ __launch_bounds__(256) void aug_bw_resp_kernel(const ChainParams p) {
  const int stream = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  const mst_aug_stem& d = p.dec[stream >> 3].stem[(stream & 7) >> 1];
  if (d.bw_sections <= 0 || c >= p.nchunk) return;
  float* x = p.stems + (size_t)stream * p.T;
  const double* st = p.states + ((size_t)stream * p.nchunk + c) * 4;
  const Coef kb0 = coef_of(d.bw_sos), kb1 = coef_of(d.bw_sos + 6);
  const bool two = d.bw_sections > 1;
  double b0 = st[0], b1 = st[1], b2 = two ? st[2] : 0.0, b3 = two ? st[3] : 0.0;
  walk_chunk<true>(x, c * kLc, min(p.T, (c + 1) * kLc), [&](float v) {
    double w = biquad_step(kb0, b0, b1, (double)v);
    if (two) w = biquad_step(kb1, b2, b3, w);
    return (float)w;
  });
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
  bool any_tilt = false, any_bw = false;
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < 4; ++s)
      any_tilt = any_tilt || decisions[b].stem[s].tilt != 0, any_bw = any_bw || decisions[b].stem[s].bw_sections > 0;
  const dim3 cgrid((L.nchunk + 255) / 256, B * 8);
  if (any_tilt) {
    hipLaunchKernelGGL(aug_tilt_zs_kernel, cgrid, dim3(256), 0, st, cp);
    hipLaunchKernelGGL((aug_scan_kernel<0>), dim3(B * 8), dim3(64), 0, st, cp);
  }
  hipLaunchKernelGGL(aug_tilt_resp_kernel, cgrid, dim3(256), 0, st, cp);
  if (any_bw) {
    hipLaunchKernelGGL((aug_scan_kernel<1>), dim3(B * 8), dim3(64), 0, st, cp);
    hipLaunchKernelGGL(aug_bw_resp_kernel, cgrid, dim3(256), 0, st, cp);
  }
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
