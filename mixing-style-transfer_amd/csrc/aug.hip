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

#include <cstdlib>

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
  long long clip_stride;     // floats between clips (8 * T when packed)
};

// mixing_utils.py:435-447 (threshold -20 dB, ratio 4): dB = 20 log10(|x| + 1e-8); above -20 dB: dB' = -20 + (dB + 20) / 4;
// y = sign(x) 10^(dB' / 20).  In closed form, with a = |x| + 1e-8:  a <= 0.1 -> y = sign * a (the 1e-8 offset survives, as in
// the reference);  a > 0.1 -> y = sign * 10^(-3/4) * a^(1/4).  Two square roots instead of log10f + powf: ~10 instructions
// instead of ~100, and closer to the real-valued result than the reference's own float32 log / pow chain (which it matches to
// a few 1e-7 relative; the goldens hold at 1e-5).
__device__ __forceinline__ float compress_f32(float x) {
  const float a = fabsf(x) + 1e-8f;
  const float y = a > 0.1f ? 0.17782794100389228f * sqrtf(sqrtf(a)) : a;
  return x > 0.f ? y : (x < 0.f ? -y : 0.f);
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
  float* x = p.stems + (size_t)(stream >> 3) * p.clip_stride + (size_t)(stream & 7) * p.T;
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
  float* x = p.stems + (size_t)(stream >> 3) * p.clip_stride + (size_t)(stream & 7) * p.T;
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
  float* x = p.stems + (size_t)(stream >> 3) * p.clip_stride + (size_t)(stream & 7) * p.T;
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


// ------------------------------------------------------------------------------------------
// The three streaming passes of the IIR chain on LDS slabs (the aligned case: T % 4 == 0, 16-byte aligned base).
// A thread still owns one 512-sample chunk and walks it serially in float64 -- but the samples no longer come from the
// thread's own strided 16-byte loads / stores (64 lanes -> 64 different 128-byte lines per instruction, every line fetched
// 8 times and written back in 8 partial pieces: the chain ran at 5 % of the HBM rate).  A workgroup = 256 consecutive chunks
// of one stream; per step it moves a SLAB of 32 samples of each of its chunks: 8 consecutive threads read / write one whole
// 128-byte run of a chunk, the slab is transposed through LDS ([256 chunks][33]: the thread of chunk r walks row r, stride
// 33 words = no bank conflicts), the next slab's loads are in flight while this one is computed.
//   PASS 0  zero-state pass of the tilt biquad on x * gain                                  (read only)
//   PASS 1  gain -> tilt response -> compressor -> store; low-pass zero-state pass on the stored values   (read + write)
//   PASS 2  low-pass response from the scanned start states                                   (read + write)
// Same per-sample arithmetic as the chunk-walk kernels above (which stay as the path for unaligned lengths).
// ------------------------------------------------------------------------------------------
constexpr int kSlab = 32, kSlabRow = kSlab + 1, kWgChunks = 256;

template <int PASS>
__global__ __launch_bounds__(256) void aug_iir_pass_kernel(const ChainParams p) {
  __shared__ float tile[2][kWgChunks * kSlabRow];
  const int stream = blockIdx.y, tid = threadIdx.x;
  const mst_aug_stem& d = p.dec[stream >> 3].stem[(stream & 7) >> 1];
  const bool has_g = d.gain != 1.0f, has_t = d.tilt != 0, has_c = d.compress != 0, has_b = d.bw_sections > 0;
  if (PASS == 0 && !has_t) return;                                   // block-uniform early outs
  if (PASS == 1 && !(has_g || has_t || has_c || has_b)) return;
  if (PASS == 2 && !has_b) return;
  const int c0 = blockIdx.x * kWgChunks, c = c0 + tid;              // this thread's chunk
  if (c0 >= p.nchunk) return;
  float* x = p.stems + (size_t)(stream >> 3) * p.clip_stride + (size_t)(stream & 7) * p.T;
  const bool writes = PASS == 2 || (PASS == 1 && (has_g || has_t || has_c));
  const Coef kt = coef_of(d.tilt_sos), kb0 = coef_of(d.bw_sos), kb1 = coef_of(d.bw_sos + 6);
  const float gain = d.gain;
  const bool two = d.bw_sections > 1;
  double* st = p.states + ((size_t)stream * p.nchunk + min(c, p.nchunk - 1)) * 4;
  double t0 = 0.0, t1 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
  if (PASS == 1 && has_t && c < p.nchunk) t0 = st[0], t1 = st[1];
  if (PASS == 2 && c < p.nchunk) b0 = st[0], b1 = st[1], b2 = two ? st[2] : 0.0, b3 = two ? st[3] : 0.0;
  // mover mapping: instruction i of a slab: chunk row 32 i + (tid >> 3), 16-byte part tid & 7
  const int mrow = tid >> 3, mpart = tid & 7;
  float4 reg[8];
  auto load_slab = [&](int k) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int cc = c0 + 32 * i + mrow;
      const long long n = (long long)cc * kLc + kSlab * k + 4 * mpart;
      reg[i] = (cc < p.nchunk && n + 3 < p.T) ? *reinterpret_cast<const float4*>(x + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load_slab(0);
  for (int k = 0; k < kLc / kSlab; ++k) {
    float* tl = tile[k & 1];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float* q = tl + (32 * i + mrow) * kSlabRow + 4 * mpart;
      q[0] = reg[i].x, q[1] = reg[i].y, q[2] = reg[i].z, q[3] = reg[i].w;
    }
    __syncthreads();
    if (k + 1 < kLc / kSlab) load_slab(k + 1);
    {
      float* row = tl + tid * kSlabRow;
      const int nvalid = c < p.nchunk ? max(0, min(kSlab, p.T - (c * kLc + kSlab * k))) : 0;   // samples of this slab inside the clip
      if (nvalid == kSlab) {
#pragma unroll 4
        for (int j = 0; j < kSlab; ++j) {
          const float v = row[j];
          if (PASS == 0) {
            (void)biquad_step(kt, t0, t1, (double)(v * gain));
          } else if (PASS == 1) {
            float y = v * gain;
            if (has_t) y = (float)biquad_step(kt, t0, t1, (double)y);
            if (has_c) y = compress_f32(y);
            if (has_b) {
              const double w = biquad_step(kb0, b0, b1, (double)y);
              if (two) (void)biquad_step(kb1, b2, b3, w);
            }
            row[j] = y;
          } else {
            double w = biquad_step(kb0, b0, b1, (double)v);
            if (two) w = biquad_step(kb1, b2, b3, w);
            row[j] = (float)w;
          }
        }
      } else {
        for (int j = 0; j < nvalid; ++j) {   // the clip's last, ragged chunk
          const float v = row[j];
          if (PASS == 0) {
            (void)biquad_step(kt, t0, t1, (double)(v * gain));
          } else if (PASS == 1) {
            float y = v * gain;
            if (has_t) y = (float)biquad_step(kt, t0, t1, (double)y);
            if (has_c) y = compress_f32(y);
            if (has_b) {
              const double w = biquad_step(kb0, b0, b1, (double)y);
              if (two) (void)biquad_step(kb1, b2, b3, w);
            }
            row[j] = y;
          } else {
            double w = biquad_step(kb0, b0, b1, (double)v);
            if (two) w = biquad_step(kb1, b2, b3, w);
            row[j] = (float)w;
          }
        }
      }
    }
    if (writes) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int cc = c0 + 32 * i + mrow;
        const long long n = (long long)cc * kLc + kSlab * k + 4 * mpart;
        if (cc < p.nchunk && n + 3 < p.T) {
          const float* q = tl + (32 * i + mrow) * kSlabRow + 4 * mpart;
          *reinterpret_cast<float4*>(x + n) = make_float4(q[0], q[1], q[2], q[3]);
        }
      }
    }
  }
  if (c < p.nchunk) {
    if (PASS == 0) st[0] = t0, st[1] = t1;
    if (PASS == 1 && has_b) st[0] = b0, st[1] = b1, st[2] = b2, st[3] = b3;
  }
}

// mean(stem^2) over (2, T) per stem  ->  redistribution weights E_s / (sum_s E_s + 1e-8)   (:410-416); the per-window partial
// sums come from rev_fft_kernel<1>
// one wave per clip: fixed-order (hence run-to-run identical) sum of the per-window partials, then the weights
__global__ __launch_bounds__(64) void aug_energy_final_kernel(const mst_aug_clip* dec, const double* partial, float* prop, int T,
                                                               int NX) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (dec[b].reverb != 1) return;
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = lane; i < NX; i += 64)
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] += partial[((size_t)b * NX + i) * 4 + s];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a[s] += __shfl_xor(a[s], o, 64);
  if (lane != 0) return;
  float E[4], tot = 0.f;
  for (int s = 0; s < 4; ++s) {
    E[s] = (float)(a[s] / (2.0 * T));
    tot += E[s];  // python sum([...]) : (((0 + E0) + E1) + E2) + E3
  }
  tot += 1e-8f;
  for (int s = 0; s < 4; ++s) prop[b * 4 + s] = E[s] / tot;
}

struct RevParams {
  float* stems;               // [B][8][T]
  const mst_aug_clip* dec;
  const float* ir;            // [B][L]
  float* prop;                // [B][4]
  double* epart;              // [B][NX][4] per-window energy partials of the 4 stems (written by rev_fft_kernel<1>)
  float2* G;                  // [B][NP][16][64]  spectra of the reversed-IR partitions (register order)
  float2* X;                  // [B][NX][16][64]  spectra of the input windows
  int T, L, NP, NX, D, j0, nj;
  int vec4;                   // T % 4 == 0, D % 4 == 0 and a 16-byte aligned base: 16-byte accesses in the redistribution
  long long clip_stride;      // floats between clips
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
    const size_t co = (size_t)b * p.clip_stride;
    // the window's samples of all 8 channels pass through here anyway: the per-stem energies of the redistribution weights
    // (mixing_utils.py:410-416) are summed on the way -- every sample once, in the window whose SECOND half holds it -- as one
    // double per (clip, window, stem), reduced in a fixed order by aug_energy_final_kernel
    double en[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = lane + 64 * in_q<kNfft>(r);
      const int n = kBlk * (item - 1) + t;
      const bool ok = n >= 0 && n < p.T;
      if (mode == 1) {
        float c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = ok ? p.stems[co + (size_t)k * p.T + n] : 0.f;
        v[r] = make_float2(((c[0] + c[2]) + c[4]) + c[6], ((c[1] + c[3]) + c[5]) + c[7]);   // ((v+b)+d)+o
        if (t >= kBlk) {
#pragma unroll
          for (int k = 0; k < 4; ++k) en[k] += (double)c[2 * k] * (double)c[2 * k] + (double)c[2 * k + 1] * (double)c[2 * k + 1];
        }
      } else {
        v[r] = ok ? make_float2(mix_at(p.stems, co, p.T, 0, n, mode), mix_at(p.stems, co, p.T, 1, n, mode))
                  : make_float2(0.f, 0.f);
      }
    }
    if (mode == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double a = en[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) p.epart[((size_t)b * p.NX + item) * 4 + k] = a;
      }
    }
  }
  FftPlan<kNfft>::run<1>(vv, lds.scr[wave], lds.tw, lane);
  float2* dst = (KIND == 0 ? p.G + ((size_t)b * p.NP + item) * 1024 : p.X + ((size_t)b * p.NX + item) * 1024) + lane;
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[r * 64] = v[r];
}

// Spectral multiply-accumulate over the NP partitions + inverse FFT + redistribution.  A workgroup = 8 waves = 8 consecutive
// output blocks j; wave w needs X[j0 + w - q] and G[q] at step q.  Every spectrum is 8 KB: read per wave from global memory
// that was 2 x 44 x 8 KB per output block, 7 GB through L2 per step (the kernel ran at the L2 rate).  Here the workgroup
// shares them through LDS: G[q] is loaded once per step for all 8 waves (double-buffered), and the X windows slide -- step q
// needs ONE new block, X[j0 - q], which replaces the block only step q - 1's last wave still used (a ring of 9 slots, so
// that the write of step q never touches what step q - 1 reads: one barrier per step).  Global traffic per step: 16 KB per
// workgroup instead of 128 KB.
constexpr int kRevWaves = 8, kRevRing = kRevWaves + 1;
struct RevLds {
  float2 tw[FftPlan<kNfft>::TW];
  union {
    struct {
      float2 X[kRevRing][1024];
      float2 G[2][1024];
    } r;
    float2 scr[kRevWaves][kNfft + kNfft / 8];   // the inverse FFT's scratch, after the last step
  } u;
};

__global__ __launch_bounds__(kRevWaves * 64) void rev_mac_ifft_kernel(const RevParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rev_smem[];
  RevLds& lds = *reinterpret_cast<RevLds*>(rev_smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int mode = p.dec[b].reverb;
  if (mode == 0) return;   // block-uniform
  fill_twiddles<kNfft>(lds.tw, tid, kRevWaves * 64);
  const int jb = p.j0 + blockIdx.x * kRevWaves;   // first output block of this workgroup
  const int j = jb + wave;
  const bool mine = blockIdx.x * kRevWaves + wave < p.nj;
  const float4* Gg = reinterpret_cast<const float4*>(p.G + (size_t)b * p.NP * 1024);
  const float4* Xg = reinterpret_cast<const float4*>(p.X + (size_t)b * p.NX * 1024);
  auto slot = [](int i) { return ((i % kRevRing) + kRevRing) % kRevRing; };
  // prologue: the 8 windows of step 0
  for (int w = 0; w < kRevWaves; ++w) {
    const int i = jb + w;
    const float4 v = (i >= 0 && i < p.NX) ? Xg[(size_t)i * 512 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(lds.u.r.X[slot(i)])[tid] = v;
  }
  float4 gq = Gg[tid], xq = make_float4(0.f, 0.f, 0.f, 0.f);   // G[0]; the new window of step 0 is already in place
  float2 acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = make_float2(0.f, 0.f);
  for (int q = 0; q < p.NP; ++q) {
    reinterpret_cast<float4*>(lds.u.r.G[q & 1])[tid] = gq;
    if (q > 0) reinterpret_cast<float4*>(lds.u.r.X[slot(jb - q)])[tid] = xq;
    __syncthreads();
    if (q + 1 < p.NP) {   // next step's two blocks, in flight while this step computes
      gq = Gg[(size_t)(q + 1) * 512 + tid];
      const int i = jb - (q + 1);
      xq = (i >= 0 && i < p.NX) ? Xg[(size_t)i * 512 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int i = j - q;
    if (mine && i >= 0 && i < p.NX) {   // wave-uniform
      const float2* x = lds.u.r.X[slot(i)] + lane;
      const float2* g = lds.u.r.G[q & 1] + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float2 a = x[r * 64], w = g[r * 64];
        acc[r].x = fmaf(a.x, w.x, fmaf(-a.y, w.y, acc[r].x));
        acc[r].y = fmaf(a.x, w.y, fmaf(a.y, w.x, acc[r].y));
      }
    }
  }
  __syncthreads();   // the ring becomes the FFT scratch
  if (!mine) return;
  // inverse FFT = conj(FFT(conj(Y))) / N; acc is in OUTPUT register order, the FFT wants INPUT register order
  float2 vv[1][16];
  float2 (&v)[16] = vv[0];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float2 y = acc[out_reg<kNfft>(in_q<kNfft>(r))];
    v[r] = make_float2(y.x, -y.y);
  }
  FftPlan<kNfft>::run<1>(vv, lds.u.scr[wave], lds.tw, lane);
  const float scale = 1.0f / (float)kNfft;
  const size_t co = (size_t)b * p.clip_stride;
  float pr[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) pr[s] = mode == 1 ? p.prop[b * 4 + s] : 0.f;
  const int nbase = kBlk * j - p.D;   // sample index of window position 512
  if (mode == 1 && p.vec4 && nbase >= 0 && nbase + kBlk <= p.T) {
    // interior block, aligned clip: the 512 valid outputs go through the wave's scratch so that a lane owns 4 CONSECUTIVE samples --
    // 8 loads + 8 stores of 16 bytes per group for the whole read-modify-write of the 8 stems (the strided form below
    // takes 24 dword accesses per sample and ran at a third of this kernel's time)
    float2* ys = lds.u.scr[wave];   // the FFT is done with it
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 8; q < 16; ++q) {
      const float2 z = v[out_reg<kNfft>(q)];
      ys[lane + 64 * (q - 8)] = make_float2(z.x * scale, -z.y * scale);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int o = 4 * (lane + 64 * h);
      const int n = nbase + o;
      float4 x[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] = *reinterpret_cast<const float4*>(p.stems + co + (size_t)c * p.T + n);
      float revL[4], revR[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float2 y = ys[o + e];
        const float* f = reinterpret_cast<const float*>(x);
        const float mL = ((f[0 * 4 + e] + f[2 * 4 + e]) + f[4 * 4 + e]) + f[6 * 4 + e];   // ((v+b)+d)+o
        const float mR = ((f[1 * 4 + e] + f[3 * 4 + e]) + f[5 * 4 + e]) + f[7 * 4 + e];
        revL[e] = mL * 0.7f + y.x * 0.3f, revR[e] = mR * 0.7f + y.y * 0.3f;
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float* r = (c & 1) ? revR : revL;
        const float w = pr[c >> 1];
        float4 o4;
        o4.x = x[c].x + (r[0] * w) * 0.3f, o4.y = x[c].y + (r[1] * w) * 0.3f;
        o4.z = x[c].z + (r[2] * w) * 0.3f, o4.w = x[c].w + (r[3] * w) * 0.3f;
        *reinterpret_cast<float4*>(p.stems + co + (size_t)c * p.T + n) = o4;
      }
    }
    return;
  }
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
  a.G = take((size_t)B * a.NP * 1024 * sizeof(float2));
  a.X = take((size_t)B * a.NX * 1024 * sizeof(float2));
  a.epart = take((size_t)B * a.NX * 4 * sizeof(double));
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
  return mst_aug_apply_strided(decisions, B, T, stems_inout, (long long)8 * T, reverb_ir, ir_len, workspace, workspace_bytes, stream);
}

int mst_aug_apply_strided(const mst_aug_clip* decisions, int B, int T, float* stems_inout, long long clip_stride,
                          const float* reverb_ir, int ir_len, void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(decisions && stems_inout, "mst_aug_apply: NULL argument");
  MST_REQUIRE(clip_stride >= (long long)8 * T, "mst_aug_apply_strided: clip_stride %lld < 8 * T", clip_stride);
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
  ChainParams cp{stems_inout, ddec, reinterpret_cast<double*>(ws + L.states), T, L.nchunk, clip_stride};
  bool any_tilt = false, any_bw = false;
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < 4; ++s)
      any_tilt = any_tilt || decisions[b].stem[s].tilt != 0, any_bw = any_bw || decisions[b].stem[s].bw_sections > 0;
  const dim3 cgrid((L.nchunk + 255) / 256, B * 8);
  // LDS-slab passes (whole-line loads / stores) when every stream is 16-byte aligned; the chunk-walk kernels otherwise
  const bool slabs = T % 4 == 0 && clip_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(stems_inout) & 15) == 0 &&
                     !getenv("MST_AUG_CHUNKWALK");
  if (any_tilt) {
    if (slabs) hipLaunchKernelGGL((aug_iir_pass_kernel<0>), cgrid, dim3(256), 0, st, cp);
    else hipLaunchKernelGGL(aug_tilt_zs_kernel, cgrid, dim3(256), 0, st, cp);
    hipLaunchKernelGGL((aug_scan_kernel<0>), dim3(B * 8), dim3(64), 0, st, cp);
  }
  if (slabs) hipLaunchKernelGGL((aug_iir_pass_kernel<1>), cgrid, dim3(256), 0, st, cp);
  else hipLaunchKernelGGL(aug_tilt_resp_kernel, cgrid, dim3(256), 0, st, cp);
  if (any_bw) {
    hipLaunchKernelGGL((aug_scan_kernel<1>), dim3(B * 8), dim3(64), 0, st, cp);
    if (slabs) hipLaunchKernelGGL((aug_iir_pass_kernel<2>), cgrid, dim3(256), 0, st, cp);
    else hipLaunchKernelGGL(aug_bw_resp_kernel, cgrid, dim3(256), 0, st, cp);
  }
  MST_HIP_CHECK(hipGetLastError());
  if (any_rev) {
    float* prop = reinterpret_cast<float*>(ws + L.prop);
    double* epart = reinterpret_cast<double*>(ws + L.epart);
    RevParams rp{stems_inout, ddec, reverb_ir, prop, epart, reinterpret_cast<float2*>(ws + L.G),
                 reinterpret_cast<float2*>(ws + L.X), T, ir_len, L.NP, L.NX, L.D, L.j0, L.nj,
                 (T % 4 == 0 && L.D % 4 == 0 && clip_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(stems_inout) & 15) == 0) ? 1 : 0,
                 clip_stride};
    hipLaunchKernelGGL((rev_fft_kernel<0>), dim3((L.NP + 3) / 4, B), dim3(256), 0, st, rp);
    hipLaunchKernelGGL((rev_fft_kernel<1>), dim3((L.NX + 3) / 4, B), dim3(256), 0, st, rp);   // (+ the stems' energy partials)
    hipLaunchKernelGGL(aug_energy_final_kernel, dim3(B), dim3(64), 0, st, ddec, epart, prop, T, L.NX);
    static unsigned long long rev_attr = 0;   // per-device bit mask: the dynamic-LDS limit belongs to the device
    if (mst::first_use_on_device(rev_attr))
      MST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rev_mac_ifft_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RevLds)));
    hipLaunchKernelGGL(rev_mac_ifft_kernel, dim3((L.nj + kRevWaves - 1) / kRevWaves, B), dim3(kRevWaves * 64), sizeof(RevLds), st, rp);
    MST_HIP_CHECK(hipGetLastError());
  }
  return MST_OK;
}

}  // extern "C"
