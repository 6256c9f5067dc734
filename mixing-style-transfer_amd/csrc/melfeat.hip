// Stage A: waveform -> windowed FFT -> |X|^2 -> sparse HTK mel -> log-mel + mixing-feature
// partial sums, and the finalise kernel that turns the partials into the 64-d feature vector.
//
// Replaces (reference barry-mir/mixing-style-transfer): torchaudio MelSpectrogram as called at
// src/mixing_utils.py:159,280 and src/model.py:58-65; MixingFeatureExtractor src/mixing_utils.py:71-357.
//
// Kernel shape (gfx950, wave64):
//   * one workgroup = one (clip, run of ceil(F/32) frames), all 8 channels; blockIdx is XCD-remapped so the runs of
//     one clip sit on one XCD (overlapping frame reads and partial output lines meet in one L2);
//   * melfeat_spw_kernel (standard configuration): 12 waves, 3 per stem; a wave transforms the L and R channel of
//     one frame as ONE n_fft-point complex Stockham FFT held in registers (radix-8/4 butterflies), exchanged
//     through a wave-private LDS scratch between passes; twiddles / window / sparse mel weights live in LDS tables;
//     melfeat_kernel (any hop, odd T, 256 mels, n_fft 512 / 2048): 8 waves, two frames per wave;
//   * each sample is read from HBM once (frames overlap 4x, re-reads hit L1/L2), each log-mel value is written
//     once, staged through an LDS tile so that stores are frame-contiguous;
//   * all feature statistics (per-band dB sums, flatness sums, inter-stem masking, waveform moments) are
//     accumulated in registers in the same pass and reduced per workgroup.
// The sample type is a template parameter (float, or int16 PCM converted in the load).
#include "common.h"
#include "fft_wave.h"
#include "fft_pk.h"

#include <algorithm>
#include <cmath>
#include <vector>


namespace {

constexpr int kWaves = 8;              // waves per workgroup
constexpr int kFPW = 2;                // frames per wave per batch = FFTs a wave runs side by side (ILP)
constexpr int kTF = kWaves * kFPW;     // frames per workgroup batch (one LDS tile flush)
constexpr int kTileStride = kTF + 1;   // +1: conflict-free transposed tile writes
constexpr int kThreads = kWaves * 64;
constexpr int nf_of(int n_fft) { return n_fft >= 2048 ? 1 : kFPW; }  // 2 side-by-side FFTs unless registers forbid
constexpr int kNumScalars = 80;        // scalar slots per partial record (see enum below)

// scalar slots of a partial record, after the 4*M per-band sums
enum : int {
  S_LOGSUM = 0,   // [4] sum log(mel+1e-10) over (2 ch, M, frames)
  S_LINSUM = 4,   // [4] sum mel
  S_MASK = 8,     // [4] sum sigmoid(max_other - own)
  S_SQ = 12,      // [8] sum x^2
  S_PEAK = 20,    // [8] max |x|
  S_DSUM = 28,    // [8] sum (x - pivot)
  S_DSQ = 36,     // [8] sum (x - pivot)^2
  S_PIVOT = 44,   // [8] pivot (first owned sample)
  S_CROSS = 52,   // [4] sum (L - pL)(R - pR)
  S_MID = 56,     // [4] sum (L + R)^2
  S_SIDE = 60,    // [4] sum (L - R)^2
  S_MIX = 64,     // [1] sum over both channels of (v + b + d + o)^2
  S_NSAMP = 65,   // [1] owned samples
  S_NFRAME = 66,  // [1] frames in this run
};

struct LaneBand {  // one mel band owned by a lane: band id (-1 = none) and first bin of its support
  int band, start;
};

struct KParams {
  const void* stem[4];       // per-stem base pointers (float or int16 PCM), each [B][2][T], clip stride `clip_stride` samples
  long long clip_stride;     // 8*T for a packed [B][8][T] tensor, 2*T for four separate [B][2][T] tensors
  float* logmel;
  float* partials;
  const float* window;
  const float2* tw;
  const float2* tw2;         // twiddles of the n_fft-point complex FFT (frame-pair packing, n_fft <= 1024)
  const float2* post;
  const float* melw;
  const LaneBand* lanebands;  // [NB][64]
  int B, T, F, M, hop;
  int tw_count, tw2_count, nnz;  // nnz = floats in the padded weight table melw[slot][i][lane]
  int glen[4], goff[4];      // per band slot r: uniform gather length (max support over lanes) and table offset
  int frames_per_run, runs_per_clip, pstride;
  int vec_ok;  // 8-byte aligned float2 frame loads allowed
  int vec4_ok; // 16-byte aligned float4 stats loads allowed
  int tile_bufs;  // 2: double-buffered output tile (one barrier per stem); 1: single (LDS-limited configs)
};

using namespace mstfft;

__device__ __forceinline__ int reflect_idx(int i, int T) { return i < 0 ? -i : (i >= T ? 2 * (T - 1) - i : i); }

__device__ __forceinline__ float2 shfl2(float2 a, int src) {
  return make_float2(__shfl(a.x, src, 64), __shfl(a.y, src, 64));
}

// Sample type of the waveform in HBM: fp32, or int16 PCM (the ingest format: half the PCIe / HBM bytes; converted
// with the exact scale 2^-15, so the result equals the fp32 path run on float(pcm) / 32768 bit for bit).
template <typename ST> struct Smp;
template <> struct Smp<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ float lds(const float* p) { return *p; }
  static __device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
  static __device__ __forceinline__ void ld4(const float* p, float (&x)[4]) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    x[0] = q.x, x[1] = q.y, x[2] = q.z, x[3] = q.w;
  }
};
template <> struct Smp<short> {
  static constexpr float kScale = 1.0f / 32768.0f;
  static __device__ __forceinline__ float ld(const short* p) { return (float)*p * kScale; }
  static __device__ __forceinline__ float lds(const short* p) {
    return (float)(*p) * kScale;
  }
  static __device__ __forceinline__ float2 ld2(const short* p) {
    const short2 q = *reinterpret_cast<const short2*>(p);
    return make_float2((float)q.x * kScale, (float)q.y * kScale);
  }
  static __device__ __forceinline__ void ld4(const short* p, float (&x)[4]) {
    const short4 q = *reinterpret_cast<const short4*>(p);
    x[0] = (float)q.x * kScale, x[1] = (float)q.y * kScale, x[2] = (float)q.z * kScale, x[3] = (float)q.w * kScale;
  }
};

// mel power of NF frames of one channel, computed side by side (two independent FFT dependency chains per wave;
// window, twiddles and mel weights are loaded once for both): returns this lane's NB band values per frame.
// With hop == NFFT/4 the samples a frame "owns" for the waveform moments, [f*hop, (f+1)*hop), are elements
// Q/2 .. Q/2 + Q/4 - 1 of every lane (float2 index lane + 64*q); RAW builds return them un-windowed in `raw`.
template <int NFFT, int NB, int NF, bool RAW = false, typename ST = float>
__device__ __forceinline__ void frames_mel(const KParams& p, const ST* const (&xchs)[NF], const int (&frame)[NF],
                                           int lane, const float2* s_win, const float2* s_tw, const float2* s_post,
                                           const float* s_melw, float2* scr, const int (&lb_start)[NB],
                                           float (&mel)[NF][NB], float2 (*raw)[NFFT / 512] = nullptr) {
  constexpr int NC = NFFT / 2;
  constexpr int Q = NC / 64;
  constexpr int NOWN = Q / 4;
  constexpr int SCR = NC + NC / 8;
  using Plan = FftPlan<NC>;
  constexpr int R0 = Plan::R0, RL = Plan::RL;
  constexpr int NBF0 = NC / R0 / 64, STR0 = NC / R0, NBFL = NC / RL / 64;
  float2 v[NF][Q];
  int s0[NF];
  bool interior = p.vec_ok != 0;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    s0[f] = frame[f] * p.hop - NFFT / 2;
    interior = interior && (s0[f] >= 0) && (s0[f] + NFFT <= p.T);
  }
  if (interior) {
#pragma unroll
    for (int u = 0; u < NBF0; ++u)
#pragma unroll
      for (int t = 0; t < R0; ++t) {
        const int n = lane + 64 * u + t * STR0;
        const float2 w = s_win[n];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const float2 x = Smp<ST>::ld2(xchs[f] + s0[f] + 2 * n);
          v[f][u * R0 + t] = make_float2(x.x * w.x, x.y * w.y);
          if constexpr (RAW) {
            const int q = u + t * NBF0;  // compile-time after unrolling
            if (q >= Q / 2 && q < Q / 2 + NOWN) raw[f][q - Q / 2] = x;
          }
        }
      }
  } else {
#pragma unroll
    for (int u = 0; u < NBF0; ++u)
#pragma unroll
      for (int t = 0; t < R0; ++t) {
        const int n = lane + 64 * u + t * STR0;
        const float2 w = s_win[n];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int i0 = s0[f] + 2 * n;
          const float2 x = make_float2(Smp<ST>::ld(xchs[f] + reflect_idx(i0, p.T)), Smp<ST>::ld(xchs[f] + reflect_idx(i0 + 1, p.T)));
          v[f][u * R0 + t] = make_float2(x.x * w.x, x.y * w.y);
          if constexpr (RAW) {
            const int q = u + t * NBF0;
            if (q >= Q / 2 && q < Q / 2 + NOWN) raw[f][q - Q / 2] = x;
          }
        }
      }
  }
  Plan::template run<NF>(v, scr, s_tw, lane);

  // real-FFT split: X[k] = E + W^k O with E,O from Z[k] and conj(Z[NC-k]); P = |X|^2
  const int mirror = (64 - lane) & 63;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    auto reg = [](int qq) { return (qq % NBFL) * RL + qq / NBFL; };
    const float2 w = s_post[q * 64 + lane];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const float2 a = v[f][reg(q)];
      const float2 bo = shfl2(v[f][reg(Q - 1 - q)], mirror);
      const float2 bs = v[f][reg((Q - q) % Q)];
      const float2 b = lane == 0 ? bs : bo;
      const float ex = 0.5f * (a.x + b.x), ey = 0.5f * (a.y - b.y);
      const float ox = 0.5f * (a.y + b.y), oy = -0.5f * (a.x - b.x);
      const float xr = ex + (w.x * ox - w.y * oy);
      const float xi = ey + (w.x * oy + w.y * ox);
      reinterpret_cast<float*>(scr + f * SCR)[lane + 64 * q] = xr * xr + xi * xi;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const float ny = v[f][0].x - v[f][0].y;
      reinterpret_cast<float*>(scr + f * SCR)[NC] = ny * ny;
    }
  }
  // sparse mel, branch-free: slot r of every lane walks glen[r] bins from its band's first bin; the weight table
  // is zero beyond a band's true support, the bin index is clamped so that only finite values are touched
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    float acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = 0.f;
    const float* w = s_melw + p.goff[r] + lane;
    const int n = p.glen[r];
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      const float wi = w[i * 64];
      const int k = min(lb_start[r] + i, NC);
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[f] = fmaf(wi, reinterpret_cast<const float*>(scr + f * SCR)[k], acc[f]);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) mel[f][r] = acc[f];
  }
}

// max over the wave of an unsigned value, returned in an SGPR: 4 DPP steps inside each row of 16 lanes
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then 4 readlanes.  No LDS traffic.
__device__ __forceinline__ unsigned wave_umax_uniform(unsigned m) {
  m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true));
  m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, true));
  m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xF, 0xF, true));
  m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xF, 0xF, true));
  const unsigned r0 = __builtin_amdgcn_readlane((int)m, 0), r1 = __builtin_amdgcn_readlane((int)m, 16);
  const unsigned r2 = __builtin_amdgcn_readlane((int)m, 32), r3 = __builtin_amdgcn_readlane((int)m, 48);
  return max(max(r0, r1), max(r2, r3));
}

// Two real frames as the real and imaginary part of ONE NFFT-point complex FFT:
//   z = a + i s b,  A[k] = (Z[k] + conj Z[N-k]) / 2,  s B[k] = (Z[k] - conj Z[N-k]) / (2i).
// The kernels pair the L and the R channel of the same stem and frame.  Unlike the even/odd-sample packing
// (half-length FFT + twiddled split, still used for n_fft = 2048) the split cancels nothing large: the cross-talk
// into A[k] is eps * |s B[k]| -- the SAME bin of the partner -- and s, an exact power of two, equalises the two
// frames' energies first, so each channel's error stays relative to its own level (the noise that remains is the
// complex FFT's own rounding, which any fp32 FFT of that frame carries).  An all-zero frame returns exact zeros.
// RAW: also return the un-windowed samples the frames own ([f*hop, (f+1)*hop): elements Q/2 .. Q/2+Q/4-1) as (a, b).
template <int NFFT, int NB, bool RAW, typename ST>
__device__ __forceinline__ void frames_mel_cplx(const KParams& p, const ST* __restrict__ xa,
                                                const ST* __restrict__ xb, int fA, int fB, int lane,
                                                const float* s_winf, const float2* s_tw2, const float* s_melw,
                                                float2* scr, const int (&lb_start)[NB], float (&mel)[2][NB],
                                                float2 (*raw)) {
  constexpr int NC = NFFT, Q = NC / 64, NOWN = Q / 4;
  using Plan = FftPlan<NC>;
  constexpr int R0 = Plan::R0, RL = Plan::RL;
  constexpr int NBF0 = NC / R0 / 64, STR0 = NC / R0, NBFL = NC / RL / 64;
  constexpr int PB_OFF = NFFT / 2 + 4;   // float offset of the second power spectrum inside the scratch
  float2 v[1][Q];
  const int sA = fA * p.hop - NFFT / 2, sB = fB * p.hop - NFFT / 2;
  const bool interior = sA >= 0 && sB >= 0 && sA + NFFT <= p.T && sB + NFFT <= p.T;
  float ma = 0.f, mb = 0.f;   // max |a w|, max |b w| of this lane
#pragma unroll
  for (int u = 0; u < NBF0; ++u)
#pragma unroll
    for (int t = 0; t < R0; ++t) {
      const int n = lane + 64 * u + t * STR0;
      const float w = s_winf[n];
      float a, b;
      if (interior) a = Smp<ST>::ld(xa + sA + n), b = Smp<ST>::ld(xb + sB + n);
      else a = Smp<ST>::ld(xa + reflect_idx(sA + n, p.T)), b = Smp<ST>::ld(xb + reflect_idx(sB + n, p.T));
      const float aw = a * w, bw = b * w;
      ma = fmaxf(ma, fabsf(aw)), mb = fmaxf(mb, fabsf(bw));
      v[0][u * R0 + t] = make_float2(aw, bw);
      if constexpr (RAW) {
        const int q = u + t * NBF0;  // compile-time after unrolling
        if (q >= Q / 2 && q < Q / 2 + NOWN) raw[q - Q / 2] = make_float2(a, b);
      }
    }
  // wave-uniform peak magnitudes (as raw bits; non-negative floats order like unsigned ints) -> scale exponent
  const unsigned ua = wave_umax_uniform(__float_as_uint(ma)), ub = wave_umax_uniform(__float_as_uint(mb));
  int sh = 0;
  if (ua != 0u && ub != 0u) sh = max(-40, min(40, (int)(ua >> 23) - (int)(ub >> 23)));
  if (sh != 0) {   // scalar branch
    const float sc = __uint_as_float((unsigned)(127 + sh) << 23);
#pragma unroll
    for (int q = 0; q < Q; ++q) v[0][q].y *= sc;
  }
  const float ia = ua != 0u ? 1.0f : 0.f;
  const float ib = ub != 0u ? __uint_as_float((unsigned)(127 - 2 * sh) << 23) : 0.f;
  Plan::template run<1>(v, scr, s_tw2, lane);
  float* PA = reinterpret_cast<float*>(scr);
  float* PB = PA + PB_OFF;
  const int mirror = (64 - lane) & 63;
  auto reg = [](int qq) { return (qq % NBFL) * RL + qq / NBFL; };
#pragma unroll
  for (int q = 0; q < Q / 2; ++q) {
    const float2 a = v[0][reg(q)];
    const float2 bo = shfl2(v[0][reg(Q - 1 - q)], mirror);
    const float2 bs = v[0][reg((Q - q) % Q)];
    const float2 b = lane == 0 ? bs : bo;
    const float ar = 0.5f * (a.x + b.x), ai = 0.5f * (a.y - b.y);   // A[k]
    const float br = 0.5f * (a.y + b.y), bi = -0.5f * (a.x - b.x);  // s B[k]
    PA[lane + 64 * q] = (ar * ar + ai * ai) * ia;
    PB[lane + 64 * q] = (br * br + bi * bi) * ib;
  }
  if (lane == 0) {
    const float2 z = v[0][reg(Q / 2)];   // bin NFFT/2: Z = A + i s B with both real
    PA[NFFT / 2] = z.x * z.x * ia;
    PB[NFFT / 2] = z.y * z.y * ib;
  }
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    float a0 = 0.f, a1 = 0.f;
    const float* w = s_melw + p.goff[r] + lane;
    const int n = p.glen[r];
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      const float wi = w[i * 64];
      const int k = min(lb_start[r] + i, NFFT / 2);
      a0 = fmaf(wi, PA[k], a0);
      a1 = fmaf(wi, PB[k], a1);
    }
    mel[0][r] = a0;
    mel[1][r] = a1;
  }
}

template <int NFFT, int NB, typename ST>
__global__ __launch_bounds__(kThreads) void melfeat_kernel(const KParams p) {
  constexpr int NC = NFFT / 2;
  constexpr int NF = nf_of(NFFT);            // FFTs a wave runs side by side
  constexpr int SCR = NF * (NC + NC / 8);    // padded float2 per wave
  constexpr float kLn2 = 0.69314718055994530942f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool PAIR = NFFT <= 1024;        // frame-pair packed full-length FFT (see frames_mel_pair)
  const int twn = PAIR ? p.tw2_count : p.tw_count;
  float2* s_win = reinterpret_cast<float2*>(smem);           // [NC] float2 = window pairs = [NFFT] floats
  float2* s_tw = s_win + NC;                                   // [twn]
  float2* s_post = s_tw + twn;                                 // [NC] (even/odd packing only)
  float2* s_scr = s_post + (PAIR ? 0 : NC);                    // [kWaves][SCR]
  float* s_melw = reinterpret_cast<float*>(s_scr + kWaves * SCR);  // [nnz padded to 4]
  float* s_tile0 = s_melw + ((p.nnz + 3) & ~3);                // [2][2*M][kTileStride] double-buffered output tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int work = mst::xcd_remap(blockIdx.x, gridDim.x);
  const int clip = work / p.runs_per_clip, run = work % p.runs_per_clip;
  const int f_begin = run * p.frames_per_run;
  const int f_end = min(p.F, f_begin + p.frames_per_run);
  const int M = p.M;

  // ---- tables -> LDS
  {
    const float2* gw = reinterpret_cast<const float2*>(p.window);
    for (int i = tid; i < NC; i += kThreads) s_win[i] = gw[i];
    const float2* gt = PAIR ? p.tw2 : p.tw;
    for (int i = tid; i < twn; i += kThreads) s_tw[i] = gt[i];
    if (!PAIR)
      for (int i = tid; i < NC; i += kThreads) s_post[i] = p.post[i];
    for (int i = tid; i < p.nnz; i += kThreads) s_melw[i] = p.melw[i];
  }
  int lb_band[NB], lb_start[NB];
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const LaneBand q = p.lanebands[r * 64 + lane];
    lb_band[r] = q.band, lb_start[r] = q.start;
  }

  auto chan = [&](int c) {
    return static_cast<const ST*>(p.stem[c >> 1]) + (size_t)clip * p.clip_stride + (size_t)(c & 1) * p.T;
  };
  float* part = p.partials + ((size_t)clip * p.runs_per_clip + run) * p.pstride;
  float* red = reinterpret_cast<float*>(s_scr);  // workgroup reduction buffer (aliases FFT scratch)

  // ---- waveform moments over the samples this run owns: [f_begin*hop, f_end*hop) (last run: to T)
  {
    const int o_begin = f_begin * p.hop;
    const int o_end = (f_end == p.F) ? p.T : min(p.T, f_end * p.hop);
    const int n_own = max(0, o_end - o_begin);
    float piv[8], sq[8], pk[8], ds[8], dq[8], cr[4], mid[4], side[4], mix = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      piv[c] = n_own > 0 ? Smp<ST>::ld(chan(c) + o_begin) : 0.f;
      sq[c] = pk[c] = ds[c] = dq[c] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) cr[s] = mid[s] = side[s] = 0.f;
    // all loads of up to kIT iterations are issued before any arithmetic: the pass is latency-bound otherwise
    constexpr int kIT = 2;
    for (int base = tid * 4; base < n_own; base += kThreads * 4 * kIT) {
      float x[kIT][8][4];
#pragma unroll
      for (int it = 0; it < kIT; ++it) {
        const int i = base + it * kThreads * 4;
        const bool full = (i + 3 < n_own);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const ST* src = chan(c) + o_begin + min(i, max(n_own - 1, 0));
          if (full && p.vec4_ok) {
            Smp<ST>::ld4(src, x[it][c]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[it][c][e] = (i + e < n_own) ? Smp<ST>::ld(src + e) : piv[c];
          }
        }
      }
#pragma unroll
      for (int it = 0; it < kIT; ++it) {
        const int i = base + it * kThreads * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (i + e < n_own) {
            float mL = 0.f, mR = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const float L = x[it][2 * s][e], R = x[it][2 * s + 1][e];
              const float dL = L - piv[2 * s], dR = R - piv[2 * s + 1];
              sq[2 * s] = fmaf(L, L, sq[2 * s]);
              sq[2 * s + 1] = fmaf(R, R, sq[2 * s + 1]);
              pk[2 * s] = fmaxf(pk[2 * s], fabsf(L));
              pk[2 * s + 1] = fmaxf(pk[2 * s + 1], fabsf(R));
              ds[2 * s] += dL;
              ds[2 * s + 1] += dR;
              dq[2 * s] = fmaf(dL, dL, dq[2 * s]);
              dq[2 * s + 1] = fmaf(dR, dR, dq[2 * s + 1]);
              cr[s] = fmaf(dL, dR, cr[s]);
              const float sm = L + R, sd = L - R;
              mid[s] = fmaf(sm, sm, mid[s]);
              side[s] = fmaf(sd, sd, side[s]);
              mL += L;  // python sum(): ((v + b) + d) + o
              mR += R;
            }
            mix = fmaf(mL, mL, mix);
            mix = fmaf(mR, mR, mix);
          }
        }
      }
    }
    // workgroup reduce: 53 sums + 8 maxima
    float vals[53];
#pragma unroll
    for (int c = 0; c < 8; ++c) vals[c] = sq[c], vals[8 + c] = ds[c], vals[16 + c] = dq[c];
#pragma unroll
    for (int s = 0; s < 4; ++s) vals[24 + s] = cr[s], vals[28 + s] = mid[s], vals[32 + s] = side[s];
    vals[36] = mix;
#pragma unroll
    for (int k = 0; k < 37; ++k) {
      const float r = mst::wave_sum(vals[k]);
      if (lane == 0) red[wave * 48 + k] = r;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float r = mst::wave_max(pk[c]);
      if (lane == 0) red[wave * 48 + 37 + c] = r;
    }
    __syncthreads();
    if (tid < 45) {
      float r = red[tid];
      for (int w = 1; w < kWaves; ++w) r = tid < 37 ? r + red[w * 48 + tid] : fmaxf(r, red[w * 48 + tid]);
      int slot;
      if (tid < 8) slot = S_SQ + tid;
      else if (tid < 16) slot = S_DSUM + (tid - 8);
      else if (tid < 24) slot = S_DSQ + (tid - 16);
      else if (tid < 28) slot = S_CROSS + (tid - 24);
      else if (tid < 32) slot = S_MID + (tid - 28);
      else if (tid < 36) slot = S_SIDE + (tid - 32);
      else if (tid == 36) slot = S_MIX;
      else slot = S_PEAK + (tid - 37);
      part[4 * M + slot] = r;
    }
    if (tid < 8) part[4 * M + S_PIVOT + tid] = piv[tid];
    if (tid == 0) {
      part[4 * M + S_NSAMP] = (float)n_own;
      part[4 * M + S_NFRAME] = (float)max(0, f_end - f_begin);
    }
    __syncthreads();  // red aliases the FFT scratch; tables are loaded too
  }

  // ---- spectral pass
  float acc_db[4][NB], acc_log[4], acc_lin[4], acc_mask[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    acc_log[s] = acc_lin[s] = acc_mask[s] = 0.f;
#pragma unroll
    for (int r = 0; r < NB; ++r) acc_db[s][r] = 0.f;
  }
  float2* scr = s_scr + wave * SCR;

  for (int fb = f_begin; fb < f_end; fb += kTF) {
    float S[4][kFPW][NB];  // channel-mean mel power per stem, frame slot, band
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int f = 0; f < kFPW; ++f)
#pragma unroll
        for (int r = 0; r < NB; ++r) S[s][f][r] = 0.f;

#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
      float* s_tile = s_tile0 + (p.tile_bufs == 2 ? (s & 1) : 0) * (2 * M * kTileStride);
      int frame[kFPW];
      bool fok[kFPW];
#pragma unroll
      for (int ff = 0; ff < kFPW; ++ff) {
        const int fr = fb + wave * kFPW + ff;
        fok[ff] = fr < f_end;
        frame[ff] = fok[ff] ? fr : f_end - 1;  // past the run: recompute a valid frame, discard the result
      }
      if (fok[0]) {  // wave-uniform
        float sm[kFPW][NB];
        float melall[2][kFPW][NB];   // [channel][frame slot][band slot]
        if constexpr (PAIR) {
#pragma unroll 1
          for (int ff = 0; ff < kFPW; ++ff) {
            float m2[2][NB];
            frames_mel_cplx<NFFT, NB, false, ST>(p, chan(2 * s), chan(2 * s + 1), frame[ff], frame[ff], lane,
                                             reinterpret_cast<const float*>(s_win), s_tw, s_melw, scr, lb_start, m2,
                                             nullptr);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int f2 = 0; f2 < kFPW; ++f2) melall[c][f2][r] = (f2 == ff) ? m2[c][r] : melall[c][f2][r];
          }
        } else {
#pragma unroll 1
          for (int cf = 0; cf < 2 * kFPW; ++cf) {
            const int c = cf / kFPW, ff = cf % kFPW;
            const int one[1] = {frame[ff]};
            float m1[1][NB];
            const ST* const x1[1] = {chan(2 * s + c)};
            frames_mel<NFFT, NB, 1, false, ST>(p, x1, one, lane, s_win, s_tw, s_post, s_melw, scr, lb_start, m1);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
              for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int f2 = 0; f2 < kFPW; ++f2)
                  melall[c2][f2][r] = (c2 == c && f2 == ff) ? m1[0][r] : melall[c2][f2][r];
          }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float (&mel)[kFPW][NB] = melall[c];
#pragma unroll
          for (int ff = 0; ff < kFPW; ++ff) {
            float lsum = 0.f, msum = 0.f;
#pragma unroll
            for (int r = 0; r < NB; ++r) {
              const bool ok = lb_band[r] >= 0 && fok[ff];
              const float lm = __log2f(mel[ff][r] + 1e-10f) * kLn2;
              if (ok) {
                s_tile[(c * M + lb_band[r]) * kTileStride + wave * kFPW + ff] = lm;
                lsum += lm;
                msum += mel[ff][r];
              }
              const float lmz = ok ? lm : 0.f;
#pragma unroll
              for (int ss = 0; ss < 4; ++ss) acc_db[ss][r] += (s == ss) ? lmz : 0.f;
              sm[ff][r] = (c == 0) ? mel[ff][r] : (sm[ff][r] + mel[ff][r]) * 0.5f;  // mel.mean(dim=0) of (2, M, F)
            }
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
              acc_log[ss] += (s == ss) ? lsum : 0.f;
              acc_lin[ss] += (s == ss) ? msum : 0.f;
            }
          }
        }
#pragma unroll
        for (int ss = 0; ss < 4; ++ss)
#pragma unroll
          for (int ff = 0; ff < kFPW; ++ff)
#pragma unroll
            for (int r = 0; r < NB; ++r) S[ss][ff][r] = (s == ss) ? sm[ff][r] : S[ss][ff][r];
      }
      // one barrier per stem: the tile of stem s is complete; its flush below overlaps the FFTs of stem s+1 (other
      // buffer), and buffer (s & 1) is written again only after the barrier of stem s+1, i.e. after this flush
      __syncthreads();
      if (p.logmel) {  // flush the stem's two channels: rows (c, band), kTF consecutive frames each
        const int f = tid % kTF, frame = fb + f;
        if (frame < f_end) {
          for (int row = tid / kTF; row < 2 * M; row += kThreads / kTF) {
            const int c = row / M, band = row - c * M;
            p.logmel[(((size_t)clip * 8 + 2 * s + c) * M + band) * p.F + frame] = s_tile[row * kTileStride + f];
          }
        }
      }
      if (p.tile_bufs == 1) __syncthreads();  // single tile: the next stem overwrites it
    }
    // inter-stem masking for this wave's frames (mixing_utils.py:288-307)
#pragma unroll
    for (int ff = 0; ff < kFPW; ++ff) {
      if (fb + wave * kFPW + ff >= f_end) continue;
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        if (lb_band[r] < 0) continue;
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
          float other = -INFINITY;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j != ss) other = fmaxf(other, S[j][ff][r]);
          const float d = S[ss][ff][r] - other;
          acc_mask[ss] += __frcp_rn(1.0f + __expf(d));  // sigmoid((0 - d) / 1)
        }
      }
    }
  }

  // ---- workgroup reduction of the spectral accumulators
  // per-band sums: same lane owns the same bands in every wave
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < NB; ++r) red[((wave * 4 + s) * NB + r) * 64 + lane] = acc_db[s][r];
  float* red2 = red + kWaves * 4 * NB * 64;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float a = mst::wave_sum(acc_log[s]), b = mst::wave_sum(acc_lin[s]), c = mst::wave_sum(acc_mask[s]);
    if (lane == 0) {
      red2[wave * 12 + S_LOGSUM + s] = a;
      red2[wave * 12 + S_LINSUM + s] = b;
      red2[wave * 12 + S_MASK + s] = c;
    }
  }
  __syncthreads();
  for (int i = tid; i < 4 * NB * 64; i += kThreads) {
    const int l = i & 63, r = (i >> 6) % NB, s = i / (64 * NB);
    float v = 0.f;
    for (int w = 0; w < kWaves; ++w) v += red[((w * 4 + s) * NB + r) * 64 + l];
    const int band = p.lanebands[r * 64 + l].band;
    if (band >= 0) part[s * M + band] = v;
  }
  if (tid < 12) {
    float v = 0.f;
    for (int w = 0; w < kWaves; ++w) v += red2[w * 12 + tid];
    part[4 * M + tid] = v;
  }
}

// ------------------------------------------------------------------------------------------
// Stage A, "stem per wave pair" layout for the standard configuration (hop == n_fft/4, n_mels <= 128, aligned
// input).  Waves 2s and 2s+1 own stem s: a wave transforms the L and the R channel of ONE frame as one complex FFT
// (frames_mel_cplx; even / odd frames of a 16-frame batch), so
//   * per-channel and L/R cross moments come from the un-windowed samples the frame owns (no moment pass; consecutive
//     frames of a wave overlap by half, the partner wave covers the rest: L1/L2 hits),
//   * all 8 channels x 16 frames of log-mel land in one frame-major LDS tile and are flushed once per batch
//     (2 barriers per batch instead of 8), the inter-stem masking is evaluated from that tile,
//   * the mixture energy comes from a short float4 pass over the batch's samples (just loaded: cache-resident).
// Writes the same partial records as melfeat_kernel (same finalise kernel).
// ------------------------------------------------------------------------------------------
// WPS = waves per stem: the workgroup has 4*WPS waves and works on batches of 2*WPS... frames: with WPS = 3 (12 waves,
// 3 per SIMD, 168 VGPRs) the batch is 6 frames (2 per wave), which is what the 160 KB of LDS leave for the tile once
// twelve 9.2 KB FFT scratches are resident; WPS = 2 is the 8-wave / 16-frame layout.
template <int WPS>
struct SpwGeom {
  static constexpr int WAVES = 4 * WPS, THREADS = WAVES * 64;
  static constexpr int FPW = WPS == 2 ? 8 : 2;     // frames per wave per batch
  static constexpr int TF = WPS * FPW;             // frames per batch
};

template <int NFFT, typename ST, int WPS>
__global__ __launch_bounds__(SpwGeom<WPS>::THREADS) void melfeat_spw_kernel(const KParams p) {
  constexpr int kWaves = SpwGeom<WPS>::WAVES, kThreads = SpwGeom<WPS>::THREADS, kTF = SpwGeom<WPS>::TF;
  constexpr int NB = 2, NOWN = NFFT / 256;
  constexpr int SCR = NFFT + NFFT / 8;
  constexpr float kLn2 = 0.69314718055994530942f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_winf = reinterpret_cast<float*>(smem);                 // [NFFT]
  float2* s_tw2 = reinterpret_cast<float2*>(s_winf + NFFT);       // [tw2_count]
  float2* s_scr = s_tw2 + p.tw2_count;                            // [kWaves][SCR]
  float* s_melw = reinterpret_cast<float*>(s_scr + kWaves * SCR);
  float* s_tile = s_melw + ((p.nnz + 3) & ~3);                    // [kTF][8*M + 1]  log-mel, frame-major
  const int M = p.M, TS = 8 * M + 1;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int work = mst::xcd_remap(blockIdx.x, gridDim.x);
  const int clip = work / p.runs_per_clip, run = work % p.runs_per_clip;
  const int f_begin = run * p.frames_per_run;
  const int f_end = min(p.F, f_begin + p.frames_per_run);
  {
    for (int i = tid; i < NFFT; i += kThreads) s_winf[i] = p.window[i];
    for (int i = tid; i < p.tw2_count; i += kThreads) s_tw2[i] = p.tw2[i];
    for (int i = tid; i < p.nnz; i += kThreads) s_melw[i] = p.melw[i];
  }
  int lb_band[NB], lb_start[NB];
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const LaneBand q = p.lanebands[r * 64 + lane];
    lb_band[r] = q.band, lb_start[r] = q.start;
  }
  auto chan = [&](int c) {
    return static_cast<const ST*>(p.stem[c >> 1]) + (size_t)clip * p.clip_stride + (size_t)(c & 1) * p.T;
  };
  float* part = p.partials + ((size_t)clip * p.runs_per_clip + run) * p.pstride;
  const int stem = wave / WPS, half = wave % WPS;   // `half`: which of the stem's WPS waves
  const ST* const xs[2] = {chan(2 * stem), chan(2 * stem + 1)};
  const int o_begin = f_begin * p.hop;
  const int o_end = (f_end == p.F) ? p.T : min(p.T, f_end * p.hop);
  const int o_piv = min(o_begin, p.T - 1);
  const float pvL = Smp<ST>::ld(xs[0] + o_piv), pvR = Smp<ST>::ld(xs[1] + o_piv);
  float2* scr = s_scr + wave * SCR;

  float acc_db[NB] = {0.f, 0.f}, acc_log = 0.f, acc_lin = 0.f;
  float sqL = 0.f, sqR = 0.f, pkL = 0.f, pkR = 0.f, dsL = 0.f, dsR = 0.f, dqL = 0.f, dqR = 0.f;
  float cr = 0.f, mid = 0.f, side = 0.f;
  float mask[4] = {0.f, 0.f, 0.f, 0.f}, mixsq = 0.f;   // accumulated per thread in the tile phase
  __syncthreads();

  for (int fb = f_begin; fb < f_end; fb += kTF) {
    // ---- phase A: every wave: 8 frames of its stem, L and R as one complex FFT
#pragma unroll 1
    for (int i = 0; i < kTF / WPS; ++i) {
      const int fr = fb + WPS * i + half;
      if (fr >= f_end) break;  // wave-uniform
      float mel[2][NB];
      float2 raw[NOWN];        // (L, R) of the samples the frame owns
      frames_mel_cplx<NFFT, NB, true, ST>(p, xs[0], xs[1], fr, fr, lane, s_winf, s_tw2, s_melw, scr, lb_start, mel, raw);
      float* trow = s_tile + (fr - fb) * TS + 2 * stem * M;
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        if (lb_band[r] < 0) continue;
        const float l0 = __log2f(mel[0][r] + 1e-10f) * kLn2, l1 = __log2f(mel[1][r] + 1e-10f) * kLn2;
        trow[lb_band[r]] = l0;
        trow[M + lb_band[r]] = l1;
        acc_db[r] += l0 + l1;
        acc_log += l0 + l1;
        acc_lin += mel[0][r] + mel[1][r];
      }
      // waveform moments from the samples this frame owns
#pragma unroll
      for (int j = 0; j < NOWN; ++j) {
        const int n = fr * p.hop + lane + 64 * j;
        const bool m = n < p.T;
        const float L = m ? raw[j].x : 0.f;
        const float R = m ? raw[j].y : 0.f;
        const float dL = m ? L - pvL : 0.f, dR = m ? R - pvR : 0.f;
        sqL = fmaf(L, L, sqL), sqR = fmaf(R, R, sqR);
        pkL = fmaxf(pkL, fabsf(L)), pkR = fmaxf(pkR, fabsf(R));
        dsL += dL, dsR += dR;
        dqL = fmaf(dL, dL, dqL), dqR = fmaf(dR, dR, dqR);
        cr = fmaf(dL, dR, cr);
        mid = fmaf(L + R, L + R, mid);
        side = fmaf(L - R, L - R, side);
      }
    }
    __syncthreads();
    // ---- phase B (all threads): flush the batch's log-mel, masking, mixture energy
    const int nf = min(kTF, f_end - fb);
    if (p.logmel) {
      const int f = tid % kTF;
      if (f < nf)
        for (int row = tid / kTF; row < 8 * M; row += kThreads / kTF)   // row = ch * M + band
          p.logmel[((size_t)clip * 8 * M + row) * p.F + fb + f] = s_tile[f * TS + row];
    }
    for (int e = tid; e < nf * M; e += kThreads) {  // (frame, band) pairs; 8 channel reads, conflict-free across bands
      const int f = e / M, m = e - f * M;
      const float* t = s_tile + f * TS + m;
      float S[4];
#pragma unroll
      for (int s = 0; s < 4; ++s)   // channel-mean mel power back from the log domain (abs. error ~1e-7 * power)
        S[s] = 0.5f * ((__expf(t[(2 * s) * M]) - 1e-10f) + (__expf(t[(2 * s + 1) * M]) - 1e-10f));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float other = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j != s) other = fmaxf(other, S[j]);
        mask[s] += __frcp_rn(1.0f + __expf(S[s] - other));
      }
    }
    {
      const int b0 = fb * p.hop, b1 = (fb + nf == p.F) ? p.T : min(p.T, (fb + nf) * p.hop);
      for (int i = b0 + tid * 4; i < b1; i += kThreads * 4) {
        float x[8][4];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const ST* src = chan(c) + i;
          if (i + 3 < b1) {
            Smp<ST>::ld4(src, x[c]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[c][e] = (i + e < b1) ? Smp<ST>::ld(src + e) : 0.f;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float mL = ((x[0][e] + x[2][e]) + x[4][e]) + x[6][e];   // python sum(): ((v + b) + d) + o
          const float mR = ((x[1][e] + x[3][e]) + x[5][e]) + x[7][e];
          mixsq = fmaf(mL, mL, mixsq);
          mixsq = fmaf(mR, mR, mixsq);
        }
      }
    }
    __syncthreads();
  }

  // ---- reductions -> partial record (same slots as melfeat_kernel)
  float* red = reinterpret_cast<float*>(s_scr);   // [kWaves][NB][64] band sums, then per-wave scalars
#pragma unroll
  for (int r = 0; r < NB; ++r) red[(wave * NB + r) * 64 + lane] = acc_db[r];
  float* red2 = red + kWaves * NB * 64;           // [kWaves][24]
  {
    const float v[13] = {acc_log, acc_lin, sqL, sqR, dsL, dsR, dqL, dqR, cr, mid, side, mixsq, 0.f};
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const float r = mst::wave_sum(v[k]);
      if (lane == 0) red2[wave * 24 + k] = r;
    }
    const float a = mst::wave_max(pkL), b = mst::wave_max(pkR);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float r = mst::wave_sum(mask[s]);
      if (lane == 0) red2[wave * 24 + 14 + s] = r;
    }
    if (lane == 0) red2[wave * 24 + 12] = a, red2[wave * 24 + 13] = b;
  }
  __syncthreads();
  for (int i = tid; i < 4 * NB * 64; i += kThreads) {
    const int l = i & 63, r = (i >> 6) % NB, s = i / (64 * NB);
    const int band = p.lanebands[r * 64 + l].band;
    if (band >= 0) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WPS; ++w) v += red[((WPS * s + w) * NB + r) * 64 + l];
      part[s * M + band] = v;
    }
  }
  if (tid < 4) {  // per-stem scalars: waves WPS*s .. WPS*s + WPS-1
    const int s = tid;
    float v[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) v[k] = 0.f;
    for (int w = 0; w < WPS; ++w) {
      const float* a = red2 + (WPS * s + w) * 24;
#pragma unroll
      for (int k = 0; k < 12; ++k) v[k] += a[k];
      v[12] = fmaxf(v[12], a[12]), v[13] = fmaxf(v[13], a[13]);
    }
    float* q = part + 4 * M;
    q[S_LOGSUM + s] = v[0];
    q[S_LINSUM + s] = v[1];
    q[S_SQ + 2 * s] = v[2], q[S_SQ + 2 * s + 1] = v[3];
    q[S_DSUM + 2 * s] = v[4], q[S_DSUM + 2 * s + 1] = v[5];
    q[S_DSQ + 2 * s] = v[6], q[S_DSQ + 2 * s + 1] = v[7];
    q[S_CROSS + s] = v[8];
    q[S_MID + s] = v[9];
    q[S_SIDE + s] = v[10];
    q[S_PEAK + 2 * s] = v[12], q[S_PEAK + 2 * s + 1] = v[13];
    float mk = 0.f;
    for (int w = 0; w < kWaves; ++w) mk += red2[w * 24 + 14 + s];
    q[S_MASK + s] = mk;
    q[S_PIVOT + 2 * s] = Smp<ST>::ld(chan(2 * s) + o_piv), q[S_PIVOT + 2 * s + 1] = Smp<ST>::ld(chan(2 * s + 1) + o_piv);
  }
  if (tid == 8) {
    float mx = 0.f;
    for (int w = 0; w < kWaves; ++w) mx += red2[w * 24 + 11];
    part[4 * M + S_MIX] = mx;
    part[4 * M + S_NSAMP] = (float)max(0, o_end - o_begin);
    part[4 * M + S_NFRAME] = (float)max(0, f_end - f_begin);
  }
}

// ------------------------------------------------------------------------------------------
// Finalise: reduce the per-run partial records of one clip in double precision and emit the
// feature vector in the reference's sorted-key layout (mixing_utils.py:320-357).
// ------------------------------------------------------------------------------------------
struct FParams {
  const float* partials;
  float* feats;
  int M, F, T, runs_per_clip, pstride, detailed_bins, feat_dim;
};

__device__ double pearson_vs_index(const double* y, int n) {
  double my = 0;
  for (int i = 0; i < n; ++i) my += y[i];
  my /= n;
  const double mx = 0.5 * (n - 1);
  double sxy = 0, sxx = 0, syy = 0;
  for (int i = 0; i < n; ++i) {
    const double dx = i - mx, dy = y[i] - my;
    sxy += dx * dy, sxx += dx * dx, syy += dy * dy;
  }
  const double std_unbiased = sqrt(syy / (n - 1));
  if (std_unbiased < 1e-6) return 0.0;  // mixing_utils.py:184
  double c = sxy / sqrt(sxx * syy);
  return fmin(1.0, fmax(-1.0, c));      // torch.corrcoef clips to [-1, 1]
}

__global__ __launch_bounds__(256) void melfeat_finalize_kernel(const FParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* band = reinterpret_cast<double*>(smem);  // [4*M] mean dB per band
  double* sc = band + 4 * p.M;                      // [kNumScalars] reduced scalars
  double* chmean = sc + kNumScalars;                // [8]
  double* chm2 = chmean + 8;                        // [8]
  double* cross = chm2 + 8;                         // [4]
  const int tid = threadIdx.x, clip = blockIdx.x, M = p.M;
  const float* base = p.partials + (size_t)clip * p.runs_per_clip * p.pstride;
  const double inv_n_db = 1.0 / (2.0 * p.F);
  const double k_db = 10.0 / log(10.0);
  for (int i = tid; i < 4 * M; i += 256) {
    double s = 0;
#pragma unroll 8
    for (int r = 0; r < p.runs_per_clip; ++r) s += base[(size_t)r * p.pstride + i];
    band[i] = s * inv_n_db * k_db;  // mean over (2 ch, F) of 10*log10(mel + 1e-10)
  }
  if (tid < kNumScalars) {
    const bool is_max = tid >= S_PEAK && tid < S_PEAK + 8;
    double s = 0;
#pragma unroll 8
    for (int r = 0; r < p.runs_per_clip; ++r) {
      const double v = base[(size_t)r * p.pstride + 4 * M + tid];
      s = is_max ? fmax(s, v) : s + v;
    }
    sc[tid] = s;
  }
  // pooled mean / M2 / cross moment from pivoted per-run sums (Chan et al. pairwise update)
  if (tid < 8) {
    double n = 0, mean = 0, m2 = 0;
    for (int r = 0; r < p.runs_per_clip; ++r) {
      const float* q = base + (size_t)r * p.pstride + 4 * M;
      const double nr = q[S_NSAMP];
      if (nr <= 0) continue;
      const double sr = q[S_DSUM + tid], qr = q[S_DSQ + tid];
      const double mr = q[S_PIVOT + tid] + sr / nr, m2r = fmax(0.0, qr - sr * sr / nr);
      const double d = mr - mean, nt = n + nr;
      m2 += m2r + d * d * n * nr / nt;
      mean += d * nr / nt;
      n = nt;
    }
    chmean[tid] = mean, chm2[tid] = m2;
  } else if (tid >= 64 && tid < 68) {
    const int s = tid - 64;
    double n = 0, mL = 0, mR = 0, c = 0;
    for (int r = 0; r < p.runs_per_clip; ++r) {
      const float* q = base + (size_t)r * p.pstride + 4 * M;
      const double nr = q[S_NSAMP];
      if (nr <= 0) continue;
      const double sL = q[S_DSUM + 2 * s], sR = q[S_DSUM + 2 * s + 1];
      const double mLr = q[S_PIVOT + 2 * s] + sL / nr, mRr = q[S_PIVOT + 2 * s + 1] + sR / nr;
      const double cr = q[S_CROSS + s] - sL * sR / nr;
      const double dL = mLr - mL, dR = mRr - mR, nt = n + nr;
      c += cr + dL * dR * n * nr / nt;
      mL += dL * nr / nt, mR += dR * nr / nt;
      n = nt;
    }
    cross[s] = c;
  }
  __syncthreads();

  const int sd = p.detailed_bins > 0 ? p.detailed_bins + 2 : 5;
  const int ps = 10 + sd;  // dynamics 6 + rel_loudness 1 + spectral + stereo 3
  float* out = p.feats + (size_t)clip * p.feat_dim;
  auto put = [&](int idx, double v) {
    float f = (float)v;
    f = fminf(fmaxf(f, -100.0f), 100.0f);  // clamp; fminf/fmaxf drop NaN like the NaN->0 step below
    if (v != v) f = 0.0f;                  // NaN -> 0 (mixing_utils.py:343-348)
    out[idx] = f;
  };
  const double n = (double)p.T;
  auto loud = [&](double ms) { return -0.691 + 10.0 * log10(ms + 1e-10); };
  if (tid < 4) {
    const int s = tid;  // stem index in input order v,b,d,o ; sorted-key block order b,d,(mask),o,v
    const int blk = (s == 1) ? 0 : (s == 2) ? ps : (s == 3) ? 2 * ps + 4 : 3 * ps + 4;
    const double msL = sc[S_SQ + 2 * s] / n, msR = sc[S_SQ + 2 * s + 1] / n;
    const double rmsL = sqrt(msL), rmsR = sqrt(msR);
    const double l_stem = loud((sc[S_SQ + 2 * s] + sc[S_SQ + 2 * s + 1]) / (2 * n));
    const double l_mix = loud(sc[S_MIX] / (2 * n));
    // dynamics: rms L,R; crest L,R; loudness x2   (mixing_utils.py:107-139)
    put(blk + 0, rmsL);
    put(blk + 1, rmsR);
    put(blk + 2, 20.0 * log10(sc[S_PEAK + 2 * s] / (rmsL + 1e-8)));
    put(blk + 3, 20.0 * log10(sc[S_PEAK + 2 * s + 1] / (rmsR + 1e-8)));
    put(blk + 4, l_stem);
    put(blk + 5, l_stem);
    put(blk + 6, l_stem - l_mix);  // rel_loudness (:94-97)
    // spectral (:141-236)
    const double* e = band + s * M;
    const double cnt = 2.0 * M * p.F;
    const double flat = exp(sc[S_LOGSUM + s] / cnt) / (sc[S_LINSUM + s] / cnt + 1e-10);
    int o = blk + 7;
    if (p.detailed_bins == 0) {
      const int q = M / 4;
      double lo = 0, mi = 0, hi = 0;
      for (int i = 0; i < q; ++i) lo += e[i];
      for (int i = q; i < 3 * q; ++i) mi += e[i];
      for (int i = 3 * q; i < M; ++i) hi += e[i];
      put(o + 0, lo / q);
      put(o + 1, mi / (2 * q));
      put(o + 2, hi / (M - 3 * q));
      put(o + 3, pearson_vs_index(e, M));
      put(o + 4, flat);
    } else {
      const int nb = p.detailed_bins;
      double* curve = reinterpret_cast<double*>(smem) + 4 * M + kNumScalars + 24 + s * nb;
      for (int i = 0; i < nb; ++i) {
        if (nb >= M) { curve[i] = e[i]; continue; }
        const float scale = nb > 1 ? (float)(M - 1) / (float)(nb - 1) : 0.f;  // align_corners=True
        const float src = scale * i;
        int i0 = (int)src;
        if (i0 > M - 1) i0 = M - 1;
        const int i1 = i0 + (i0 < M - 1 ? 1 : 0);
        const double lam = src - (float)i0;
        curve[i] = (1.0 - lam) * e[i0] + lam * e[i1];
      }
      for (int i = 0; i < nb; ++i) put(o + i, curve[i]);
      put(o + nb, pearson_vs_index(curve, nb));
      put(o + nb + 1, flat);
    }
    o = blk + 7 + sd;
    // stereo (:238-268)
    put(o + 0, 20.0 * log10(rmsL / (rmsR + 1e-8)));
    put(o + 1, cross[s] / (sqrt(chm2[2 * s] * chm2[2 * s + 1]) + 1e-8));
    put(o + 2, (sc[S_SIDE + s] / (4 * n)) / (sc[S_MID + s] / (4 * n) + 1e-8));
    // masking (:270-309): block after drums
    put(2 * ps + s, sc[S_MASK + s] / ((double)M * p.F));
  }
}

template <int NFFT, int NB, typename ST>
hipError_t launch_melfeat(const KParams& kp, int grid, size_t lds, hipStream_t st) {
  {   // dynamic-LDS limit: a per-device attribute, set on every launch (a cheap driver call)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(melfeat_kernel<NFFT, NB, ST>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((melfeat_kernel<NFFT, NB, ST>), dim3(grid), dim3(kThreads), lds, st, kp);
  return hipGetLastError();
}

template <int NFFT, typename ST, int WPS>
hipError_t launch_melfeat_spw(const KParams& kp, int grid, size_t lds, hipStream_t st) {
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(melfeat_spw_kernel<NFFT, ST, WPS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((melfeat_spw_kernel<NFFT, ST, WPS>), dim3(grid), dim3(SpwGeom<WPS>::THREADS), lds, st, kp);
  return hipGetLastError();
}

template <int NFFT>
constexpr int tw_count_of() { return FftPlan<NFFT / 2>::TW; }

#include "melfeat_v2.inc"
#include "melfeat_v2_2048.inc"

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct mst_plan {
  int sr, n_fft, hop, n_mels, detailed_bins, feat_dim;
  int nc, nb, nnz, tw_count;
  int glen[4], goff[4];
  int batches_per_run;
  float* d_window = nullptr;
  float2* d_tw = nullptr;
  float2* d_tw2 = nullptr;
  int tw2_count = 0;
  float2* d_post = nullptr;
  float* d_melw = nullptr;
  LaneBand* d_lanebands = nullptr;
  // sliding-window kernel (melfeat_v2.inc): packed-FFT twiddles and the segment form of the sparse mel table
  bool v2_ok = false;
  int v2_nslot = 0, v2_glen[kV2Slots] = {0}, v2_goff[kV2Slots] = {0}, v2_segw_count = 0, v2_wps = 3, v2_maxcnt = 1;
  int2* d_v2_bandtab = nullptr;
  // n_fft 2048 / hop 512 variant (melfeat_v2_2048.inc): W_2048 combine twiddles and the piece form of the mel table
  bool v4_ok = false;
  int v4_nslot = 0, v4_glen[4] = {0, 0, 0, 0}, v4_goff[4] = {0, 0, 0, 0}, v4_segw_count = 0, v4_maxcnt = 0;
  float2* d_v4_tw4 = nullptr;
  float2* d_v4_segw = nullptr;
  int* d_v4_pstart = nullptr;
  int* d_v4_pid = nullptr;
  int* d_v4_bandtab = nullptr;
  float2* d_v2_tw2 = nullptr;
  float2* d_v2_tw3 = nullptr;
  float2* d_v2_segw = nullptr;
  int* d_v2_segstart = nullptr;
  int* d_v2_segid = nullptr;
};

namespace {

void add_pass_tw(std::vector<float2>& tw, int NC, int R, int NS) {
  if (NS == 1) return;
  const int nbf = NS <= 64 ? 1 : NC / R / 64;  // (lane + 64 u) % NS does not depend on u when NS divides 64
  for (int u = 0; u < nbf; ++u)
    for (int t = 1; t < R; ++t)
      for (int lane = 0; lane < 64; ++lane) {
        const int j = lane + 64 * u, k = j % NS;
        const double a = -2.0 * M_PI * (double)(k * t) / (double)(NS * R);
        tw.push_back(make_float2((float)cos(a), (float)sin(a)));
      }
}

std::vector<std::pair<int, int>> passes_of(int NC) {
  switch (NC) {
    case 256: return {{4, 1}, {4, 4}, {4, 16}, {4, 64}};
    case 512: return {{8, 1}, {8, 8}, {8, 64}};
    case 1024: return {{8, 1}, {8, 8}, {4, 64}, {4, 256}};
  }
  return {};
}

// Frames per workgroup ("run").  Every workgroup costs about (frames + 2) frame-times and the device executes them in
// rounds of one workgroup per CU; with 32 runs per clip any batch that is a multiple of 8 clips fills whole rounds of
// the 256 CUs: 72 clips of 1723 frames -> 54-frame runs, 2304 workgroups = 9.0 rounds (486 frame-times per CU; fixed
// 32-frame runs took 16 rounds = 512).  The run length depends on the clip length only -- never on the batch size --
// so a clip's partial sums, and with them its features, are bit-identical whatever its batch neighbours are.
// MST_MELFEAT_BATCHES=n forces n*16-frame runs (used by the probes).
int frames_per_run_of(const mst_plan* p, int /*B*/, int F) {
  if (p->batches_per_run > 0) return p->batches_per_run * kTF;
  int fpr = (F + 31) / 32;
  fpr += fpr & 1;
  return std::max(2 * kTF, fpr);
}
int runs_per_clip(const mst_plan* p, int B, int F) {
  const int fpr = frames_per_run_of(p, B, F);
  return (F + fpr - 1) / fpr;
}
int pstride_of(const mst_plan* p) { return 4 * p->n_mels + kNumScalars; }

// sliding-window kernel: frames per wave and runs per clip (a run is WPS * fpw frames; like frames_per_run_of, a
// function of the clip length only, never of the batch size)
int v2_fpw(const mst_plan* p, int F) { return std::max(1, ((F + 31) / 32 + p->v2_wps - 1) / p->v2_wps); }
int v2_runs(const mst_plan* p, int F) {
  const int fpr = p->v2_wps * v2_fpw(p, F);
  return (F + fpr - 1) / fpr;
}

// seg(k), falling weight wl(k) (band seg - 1) and rising weight wh(k) (band seg) of every bin; false unless every bin feeds
// at most two ADJACENT bands and the segments are contiguous bin ranges.
bool segment_bins(const float* fb, int n_bins, int M, std::vector<int>& seg, std::vector<float>& wl, std::vector<float>& wh) {
  std::vector<int> peak(M, -1);
  for (int m = 0; m < M; ++m) {
    float best = 0.f;
    for (int k = 0; k < n_bins; ++k)
      if (fb[(size_t)k * M + m] > best) best = fb[(size_t)k * M + m], peak[m] = k;
  }
  seg.assign(n_bins, 0), wl.assign(n_bins, 0.f), wh.assign(n_bins, 0.f);
  int prev = 0;
  for (int k = 0; k < n_bins; ++k) {
    int b0 = -1, b1 = -1, cnt = 0;
    for (int m = 0; m < M; ++m)
      if (fb[(size_t)k * M + m] != 0.0f) {
        if (cnt == 0) b0 = m;
        b1 = m;
        ++cnt;
      }
    if (cnt > 2 || (cnt == 2 && b1 != b0 + 1)) return false;
    int s = prev;
    if (cnt == 2) {
      s = b1, wl[k] = fb[(size_t)k * M + b0], wh[k] = fb[(size_t)k * M + b1];
    } else if (cnt == 1) {
      if (k <= peak[b0]) s = b0, wh[k] = fb[(size_t)k * M + b0];
      else s = b0 + 1, wl[k] = fb[(size_t)k * M + b0];
    }
    if (s < prev) return false;
    seg[k] = prev = s;
  }
  return true;
}

// Piece form for melfeat_v2_2048_kernel: segments cut into pieces of at most 16 bins (numbered in segment order, so the
// pieces of a segment are consecutive), pieces dealt to (slot, lane) by length, and per band the piece ranges of its
// rising segment (b) and of its falling segment (b + 1).
bool v4_build_pieces(mst_plan* p, const float* fb, std::vector<float2>& segw, std::vector<int>& pstart, std::vector<int>& pid,
                     std::vector<int>& bandtab) {
  const int M = p->n_mels, n_bins = p->n_fft / 2 + 1;
  if (p->n_fft != 2048 || M > 128) return false;
  std::vector<int> seg;
  std::vector<float> wl, wh;
  if (!segment_bins(fb, n_bins, M, seg, wl, wh)) return false;
  const int nseg = M + 1;
  std::vector<int> start(nseg, 0), len(nseg, 0);
  for (int k = n_bins - 1; k >= 0; --k) start[seg[k]] = k, ++len[seg[k]];
  struct Piece { int start, len; };
  std::vector<Piece> pieces;
  std::vector<int> first(nseg, 0), cnt(nseg, 0);
  for (int sg = 0; sg < nseg; ++sg) {
    first[sg] = (int)pieces.size();
    for (int o = 0; o < len[sg]; o += 16) pieces.push_back({start[sg] + o, std::min(16, len[sg] - o)}), ++cnt[sg];
  }
  const int np = (int)pieces.size();
  if (np > 135) return false;   // exchange arrays hold 136 entries, the last one is the dump slot
  std::vector<int> order(np);
  for (int i = 0; i < np; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pieces[a].len > pieces[b].len; });
  p->v4_nslot = (np + 63) / 64;
  if (p->v4_nslot > 4) return false;
  pstart.assign((size_t)p->v4_nslot * 64, 0);
  pid.assign((size_t)p->v4_nslot * 64, 135);
  int off = 0;
  for (int r = 0; r < p->v4_nslot; ++r) {
    int gl = 0;
    for (int lane = 0; lane < 64 && r * 64 + lane < np; ++lane) gl = std::max(gl, pieces[order[r * 64 + lane]].len);
    p->v4_glen[r] = gl, p->v4_goff[r] = off;
    segw.resize((size_t)(off + gl) * 64, make_float2(0.f, 0.f));
    for (int lane = 0; lane < 64 && r * 64 + lane < np; ++lane) {
      const int pc = order[r * 64 + lane];
      pstart[(size_t)r * 64 + lane] = pieces[pc].start, pid[(size_t)r * 64 + lane] = pc;
      for (int i = 0; i < pieces[pc].len; ++i)
        segw[(size_t)(off + i) * 64 + lane] = make_float2(wl[pieces[pc].start + i], wh[pieces[pc].start + i]);
    }
    off += gl;
  }
  if (segw.size() & 1) segw.push_back(make_float2(0.f, 0.f));
  if (segw.empty()) segw.resize(2, make_float2(0.f, 0.f));
  p->v4_segw_count = (int)segw.size();
  bandtab.assign(128, 0);
  p->v4_maxcnt = 0;
  for (int b = 0; b < M; ++b) {
    bandtab[(b >> 6) * 64 + (b & 63)] = first[b] | (cnt[b] << 8) | (first[b + 1] << 16) | (cnt[b + 1] << 24);
    p->v4_maxcnt = std::max(p->v4_maxcnt, std::max(cnt[b], cnt[b + 1]));
  }
  return true;
}
int v4_fpw(int F) { return std::max(1, ((F + 31) / 32 + 1) / 2); }
int v4_runs(int F) {
  const int fpr = 2 * v4_fpw(F);
  return (F + fpr - 1) / fpr;
}

// Piece form of the triangular filterbank for melfeat_v2_kernel (n_fft 1024).  Every bin k feeds at most two ADJACENT
// bands: the falling edge of band s-1 and the rising edge of band s, where s = seg(k) numbers the interval between two
// consecutive band centres.  Segments are cut into pieces of at most 16 bins (an empty segment is one empty piece, so
// that piece index == segment index whenever no segment is longer than 16 bins -- 128 mels: the kernel's fast path),
// pieces are dealt to (slot, lane) by length, and every band gets the piece ranges of its rising segment (b) and of its
// falling segment (b + 1).  Returns false (kernel not applicable) unless the table has exactly that structure.
bool v2_build_segments(mst_plan* p, const float* fb, std::vector<float2>& segw, std::vector<int>& segstart,
                       std::vector<int>& segid, std::vector<int2>& bandtab) {
  const int M = p->n_mels, n_bins = p->n_fft / 2 + 1;
  if (p->n_fft != 1024 || M > 256) return false;
  std::vector<int> seg;
  std::vector<float> wl, wh;
  if (!segment_bins(fb, n_bins, M, seg, wl, wh)) return false;
  const int nseg = M + 1;
  std::vector<int> start(nseg, 0), len(nseg, 0);
  for (int k = n_bins - 1; k >= 0; --k) start[seg[k]] = k, ++len[seg[k]];
  struct Piece { int start, len; };
  std::vector<Piece> pieces;
  std::vector<int> first(nseg, 0), cnt(nseg, 0);
  p->v2_maxcnt = 1;
  for (int sg = 0; sg < nseg; ++sg) {
    first[sg] = (int)pieces.size();
    if (len[sg] == 0) pieces.push_back({0, 0}), cnt[sg] = 1;
    for (int o = 0; o < len[sg]; o += 16) pieces.push_back({start[sg] + o, std::min(16, len[sg] - o)}), ++cnt[sg];
    p->v2_maxcnt = std::max(p->v2_maxcnt, cnt[sg]);
  }
  const int np = (int)pieces.size();
  if (np > v2::kExDump) return false;
  std::vector<int> order(np);
  for (int i = 0; i < np; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pieces[a].len > pieces[b].len; });
  p->v2_nslot = (np + 63) / 64;
  if (p->v2_nslot > kV2Slots || (M <= 128 && p->v2_nslot > 3)) return false;
  segstart.assign((size_t)p->v2_nslot * 64, 0);
  segid.assign((size_t)p->v2_nslot * 64, v2::kExDump);
  int off = 0;
  for (int r = 0; r < p->v2_nslot; ++r) {
    int gl = 0;
    for (int lane = 0; lane < 64 && r * 64 + lane < np; ++lane) gl = std::max(gl, pieces[order[r * 64 + lane]].len);
    p->v2_glen[r] = gl, p->v2_goff[r] = off;
    segw.resize((size_t)(off + gl) * 64, make_float2(0.f, 0.f));
    for (int lane = 0; lane < 64 && r * 64 + lane < np; ++lane) {
      const int pc = order[r * 64 + lane];
      segstart[(size_t)r * 64 + lane] = pieces[pc].start, segid[(size_t)r * 64 + lane] = pc;
      for (int i = 0; i < pieces[pc].len; ++i)
        segw[(size_t)(off + i) * 64 + lane] = make_float2(wl[pieces[pc].start + i], wh[pieces[pc].start + i]);
    }
    off += gl;
  }
  if (segw.size() & 1) segw.push_back(make_float2(0.f, 0.f));
  if (segw.empty()) segw.resize(2, make_float2(0.f, 0.f));
  p->v2_segw_count = (int)segw.size();
  bandtab.assign(256, make_int2(v2::kExDump, v2::kExDump));
  for (int b = 0; b < M; ++b)
    bandtab[(size_t)(b >> 6) * 64 + (b & 63)] = make_int2(first[b] | (cnt[b] << 16), first[b + 1] | (cnt[b + 1] << 16));
  return true;
}

// the piece table has the shape melfeat_v2_kernel<..., STD> hard-codes (the 128-mel HTK filterbank at 44.1 kHz / n_fft 1024)
bool v2_table_is_standard(const mst_plan* p) {
  return p->n_mels == 128 && p->v2_nslot == 3 && p->v2_maxcnt == 1 && p->v2_glen[0] == 14 && p->v2_glen[1] == 3 && p->v2_glen[2] == 0 &&
         p->v2_goff[0] == 0 && p->v2_goff[1] == 14 && p->v2_goff[2] == 17;
}

}  // namespace

extern "C" {

int mst_plan_create(mst_plan** out, int sample_rate, int n_fft, int hop, int n_mels, const float* window,
                    const float* fb, int detailed_bins) {
  MST_REQUIRE(out && window && fb, "mst_plan_create: NULL argument");
  *out = nullptr;
  MST_REQUIRE(n_fft == 512 || n_fft == 1024 || n_fft == 2048,
              "mst_plan_create: n_fft=%d unsupported (512, 1024, 2048)", n_fft);
  MST_REQUIRE(hop >= 1 && n_mels >= 4 && n_mels <= 256, "mst_plan_create: bad hop=%d / n_mels=%d", hop, n_mels);
  MST_REQUIRE(detailed_bins >= 0 && detailed_bins <= 256 && (detailed_bins == 0 || detailed_bins < n_mels ||
              detailed_bins == n_mels), "mst_plan_create: bad detailed_bins=%d", detailed_bins);
  mst_plan* p = new mst_plan();
  p->sr = sample_rate, p->n_fft = n_fft, p->hop = hop, p->n_mels = n_mels, p->detailed_bins = detailed_bins;
  const int sd = detailed_bins > 0 ? detailed_bins + 2 : 5;
  p->feat_dim = 4 * (10 + sd) + 4;
  p->nc = n_fft / 2;
  p->nb = n_mels <= 128 ? 2 : 4;
  const char* env = getenv("MST_MELFEAT_BATCHES");
  p->batches_per_run = env ? std::max(1, atoi(env)) : 0;   // 0: chosen per launch (frames_per_run_of)
  const int n_bins = n_fft / 2 + 1;

  // sparse mel table: per band contiguous support [start, start+len)
  std::vector<int> start(n_mels, 0), len(n_mels, 0);
  for (int m = 0; m < n_mels; ++m) {
    int lo = -1, hi = -1;
    for (int k = 0; k < n_bins; ++k)
      if (fb[(size_t)k * n_mels + m] != 0.0f) {
        if (lo < 0) lo = k;
        hi = k;
      }
    if (lo >= 0) start[m] = lo, len[m] = hi - lo + 1;
  }
  // lane -> bands: slot r even: r/2*128 + lane ; r odd: (r/2)*128 + 127 - lane  (pairs a narrow low band with a
  // wide high band).  Weights go into a zero-padded table melw[goff[r] + i*64 + lane], i < glen[r] = the longest
  // support in slot r, so the gather loop has a uniform trip count and no divergence.
  std::vector<LaneBand> lbs((size_t)p->nb * 64);
  std::vector<float> melw;
  for (int r = 0; r < 4; ++r) p->glen[r] = p->goff[r] = 0;
  for (int r = 0; r < p->nb; ++r) {
    int gl = 0;
    for (int lane = 0; lane < 64; ++lane) {
      const int m = (r / 2) * 128 + ((r & 1) ? 127 - lane : lane);
      lbs[(size_t)r * 64 + lane] = (m < n_mels) ? LaneBand{m, start[m]} : LaneBand{-1, 0};
      if (m < n_mels) gl = std::max(gl, len[m]);
    }
    p->glen[r] = gl;
    p->goff[r] = (int)melw.size();
    melw.resize(melw.size() + (size_t)gl * 64, 0.f);
    for (int lane = 0; lane < 64; ++lane) {
      const int m = lbs[(size_t)r * 64 + lane].band;
      if (m < 0) continue;
      for (int i = 0; i < len[m]; ++i)
        melw[(size_t)p->goff[r] + (size_t)i * 64 + lane] = fb[(size_t)(start[m] + i) * n_mels + m];
    }
  }
  p->nnz = (int)melw.size();
  if (melw.empty()) melw.push_back(0.f);
  std::vector<float2> tw;
  for (auto pr : passes_of(p->nc)) add_pass_tw(tw, p->nc, pr.first, pr.second);
  p->tw_count = (int)tw.size();
  std::vector<float2> tw2;
  if (n_fft <= 1024)
    for (auto pr : passes_of(n_fft)) add_pass_tw(tw2, n_fft, pr.first, pr.second);
  p->tw2_count = (int)tw2.size();
  if (tw2.empty()) tw2.push_back(make_float2(1.f, 0.f));
  std::vector<float2> post(p->nc);
  for (int q = 0; q < p->nc / 64; ++q)
    for (int lane = 0; lane < 64; ++lane) {
      const int k = lane + 64 * q;
      const double a = -2.0 * M_PI * (double)k / (double)n_fft;
      post[(size_t)q * 64 + lane] = make_float2((float)cos(a), (float)sin(a));
    }
  {   // sliding-window kernel tables (standard configuration only)
    std::vector<float2> segw, t2(mstpk::kTw2Rows * 64), t3(mstpk::kTw3Rows * 64);
    std::vector<int> segstart, segid;
    std::vector<int2> bandtab;
    const char* w = getenv("MST_V2_WPS");
    p->v2_wps = ((w && atoi(w) == 2) || n_mels > 128) ? 2 : 3;   // 4 bands per lane need the 256-register build
    p->v2_ok = n_fft == 1024 && hop == 256 && v2_build_segments(p, fb, segw, segstart, segid, bandtab);
    if (p->v2_ok) {
      mstpk::fill_twiddles_host(t2.data(), t3.data());
      int rc2;
      if ((rc2 = mst::upload(&p->d_v2_tw2, t2.data(), t2.size())) || (rc2 = mst::upload(&p->d_v2_tw3, t3.data(), t3.size())) ||
          (rc2 = mst::upload(&p->d_v2_segw, segw.data(), segw.size())) ||
          (rc2 = mst::upload(&p->d_v2_segstart, segstart.data(), segstart.size())) ||
          (rc2 = mst::upload(&p->d_v2_segid, segid.data(), segid.size())) ||
          (rc2 = mst::upload(&p->d_v2_bandtab, bandtab.data(), bandtab.size()))) {
        mst_plan_destroy(p);
        return rc2;
      }
    }
  }
  {   // n_fft 2048 / hop 512 variant
    std::vector<float2> segw, t2(mstpk::kTw2Rows * 64), t3(mstpk::kTw3Rows * 64), t4(16 * 64);
    std::vector<int> pstart, pid, bandtab;
    p->v4_ok = n_fft == 2048 && hop == 512 && v4_build_pieces(p, fb, segw, pstart, pid, bandtab);
    if (p->v4_ok) {
      mstpk::fill_twiddles_host(t2.data(), t3.data());
      for (int t = 0; t < 8; ++t)
        for (int lane = 0; lane < 64; ++lane)
          for (int u = 0; u < 2; ++u) {
            const int k = (u ? (lane ? 128 - lane : 64) : lane) + 128 * t;
            const double a = -2.0 * M_PI * (double)k / 2048.0;
            t4[(size_t)(8 * u + t) * 64 + lane] = make_float2((float)cos(a), (float)sin(a));
          }
      int rc2;
      if ((rc2 = mst::upload(&p->d_v2_tw2, t2.data(), t2.size())) || (rc2 = mst::upload(&p->d_v2_tw3, t3.data(), t3.size())) ||
          (rc2 = mst::upload(&p->d_v4_tw4, t4.data(), t4.size())) || (rc2 = mst::upload(&p->d_v4_segw, segw.data(), segw.size())) ||
          (rc2 = mst::upload(&p->d_v4_pstart, pstart.data(), pstart.size())) ||
          (rc2 = mst::upload(&p->d_v4_pid, pid.data(), pid.size())) ||
          (rc2 = mst::upload(&p->d_v4_bandtab, bandtab.data(), bandtab.size()))) {
        mst_plan_destroy(p);
        return rc2;
      }
    }
  }
  int rc;
  if ((rc = mst::upload(&p->d_window, window, (size_t)n_fft)) || (rc = mst::upload(&p->d_tw, tw.data(), tw.size())) ||
      (rc = mst::upload(&p->d_tw2, tw2.data(), tw2.size())) ||
      (rc = mst::upload(&p->d_post, post.data(), post.size())) ||
      (rc = mst::upload(&p->d_melw, melw.data(), melw.size())) ||
      (rc = mst::upload(&p->d_lanebands, lbs.data(), lbs.size()))) {
    mst_plan_destroy(p);
    return rc;
  }
  *out = p;
  return MST_OK;
}

void mst_plan_destroy(mst_plan* p) {
  if (!p) return;
  (void)hipFree(p->d_window), (void)hipFree(p->d_tw), (void)hipFree(p->d_tw2), (void)hipFree(p->d_post), (void)hipFree(p->d_melw),
      (void)hipFree(p->d_lanebands);
  (void)hipFree(p->d_v2_tw2), (void)hipFree(p->d_v2_tw3), (void)hipFree(p->d_v2_segw), (void)hipFree(p->d_v2_segstart),
      (void)hipFree(p->d_v2_segid), (void)hipFree(p->d_v2_bandtab);
  (void)hipFree(p->d_v4_tw4), (void)hipFree(p->d_v4_segw), (void)hipFree(p->d_v4_pstart), (void)hipFree(p->d_v4_pid),
      (void)hipFree(p->d_v4_bandtab);
  delete p;
}

int mst_plan_frames(const mst_plan* p, int T) { return p ? 1 + T / p->hop : MST_EINVAL; }
int mst_plan_feature_dim(const mst_plan* p) { return p ? p->feat_dim : MST_EINVAL; }

size_t mst_melfeat_workspace_bytes(const mst_plan* p, int B, int T) {
  if (!p || B <= 0 || T <= 0) return 0;
  const int F = 1 + T / p->hop;
  const int runs = std::max(runs_per_clip(p, B, F), p->v2_ok ? v2_runs(p, F) : (p->v4_ok ? v4_runs(F) : 0));
  return mst::align_up((size_t)B * runs * pstride_of(p) * sizeof(float), 256);
}

}  // extern "C"

namespace {
struct LogmelOut {   // where and how the log-mel leaves stage A (mst.h: MST_LOGMEL_*)
  int layout = MST_LOGMEL_REF;
  void* lo = nullptr;          // MST_LOGMEL_CM16: low parts
  unsigned* absmax = nullptr;  // optional [B]
};
int melfeat_forward_impl(const mst_plan* p, const void* const stems4[4], bool pcm16, long long clip_stride, int B, int T,
                         float* logmel, float* feats, void* workspace, size_t workspace_bytes, void* stream,
                         const LogmelOut& lo = LogmelOut());
bool plan_uses_v2(const mst_plan* p);
}

extern "C" {

int mst_melfeat_forward(const mst_plan* p, const float* stems, int B, int T, float* logmel, float* feats,
                        void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(p && stems, "mst_melfeat_forward: NULL plan/stems");
  const void* four[4] = {stems, stems + 2 * (size_t)T, stems + 4 * (size_t)T, stems + 6 * (size_t)T};
  return melfeat_forward_impl(p, four, false, (long long)8 * T, B, T, logmel, feats, workspace, workspace_bytes, stream);
}

int mst_melfeat_forward_stems(const mst_plan* p, const float* const stems4[4], long long clip_stride, int B, int T,
                              float* logmel, float* feats, void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(p && stems4, "mst_melfeat_forward: NULL plan/stems");
  const void* four[4] = {stems4[0], stems4[1], stems4[2], stems4[3]};
  return melfeat_forward_impl(p, four, false, clip_stride, B, T, logmel, feats, workspace, workspace_bytes, stream);
}

int mst_melfeat_forward_pcm16(const mst_plan* p, const int16_t* stems, int B, int T, float* logmel, float* feats,
                              void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(p && stems, "mst_melfeat_forward_pcm16: NULL plan/stems");
  const void* four[4] = {stems, stems + 2 * (size_t)T, stems + 4 * (size_t)T, stems + 6 * (size_t)T};
  return melfeat_forward_impl(p, four, true, (long long)8 * T, B, T, logmel, feats, workspace, workspace_bytes, stream);
}

int mst_melfeat_forward_stems_pcm16(const mst_plan* p, const int16_t* const stems4[4], long long clip_stride, int B,
                                    int T, float* logmel, float* feats, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  MST_REQUIRE(p && stems4, "mst_melfeat_forward_pcm16: NULL plan/stems");
  const void* four[4] = {stems4[0], stems4[1], stems4[2], stems4[3]};
  return melfeat_forward_impl(p, four, true, clip_stride, B, T, logmel, feats, workspace, workspace_bytes, stream);
}

int mst_plan_layout_supported(const mst_plan* p, int layout) {
  if (!p) return 0;
  if (layout == MST_LOGMEL_REF) return 1;
  return (layout == MST_LOGMEL_CM32 || layout == MST_LOGMEL_CM16) && plan_uses_v2(p) ? 1 : 0;
}

int mst_melfeat_forward_io(const mst_plan* p, const mst_melfeat_io* io, int B, int T, void* workspace, size_t workspace_bytes,
                           void* stream) {
  MST_REQUIRE(p && io, "mst_melfeat_forward_io: NULL plan/io");
  MST_REQUIRE(io->layout >= MST_LOGMEL_REF && io->layout <= MST_LOGMEL_CM16, "mst_melfeat_forward_io: unknown layout %d", io->layout);
  MST_REQUIRE(mst_plan_layout_supported(p, io->layout),
              "mst_melfeat_forward_io: layout %d is not available for this plan (n_fft %d, hop %d: query mst_plan_layout_supported)",
              io->layout, p->n_fft, p->hop);
  MST_REQUIRE(io->layout == MST_LOGMEL_REF || !io->logmel || (reinterpret_cast<uintptr_t>(io->logmel) & 15) == 0,
              "mst_melfeat_forward_io: channel-minor log-mel must be 16-byte aligned");
  MST_REQUIRE(io->layout != MST_LOGMEL_CM16 || !io->logmel_lo || (reinterpret_cast<uintptr_t>(io->logmel_lo) & 15) == 0,
              "mst_melfeat_forward_io: channel-minor log-mel must be 16-byte aligned");
  LogmelOut lo;
  lo.layout = io->layout, lo.lo = io->logmel_lo, lo.absmax = io->absmax;
  const void* four[4] = {io->stems4[0], io->stems4[1], io->stems4[2], io->stems4[3]};
  return melfeat_forward_impl(p, four, io->pcm16 != 0, io->clip_stride, B, T, static_cast<float*>(io->logmel), io->feats, workspace,
                              workspace_bytes, stream, lo);
}

}  // extern "C"

namespace {

bool stage_a_env_old() {
  const char* which = getenv("MST_STAGE_A");
  return (which && (!strcmp(which, "spw") || !strcmp(which, "generic"))) || getenv("MST_MELFEAT_GENERIC");
}
bool plan_uses_v2(const mst_plan* p) { return (p->v2_ok || p->v4_ok) && !stage_a_env_old(); }

int melfeat_forward_impl(const mst_plan* p, const void* const stems4[4], bool pcm16, long long clip_stride, int B, int T,
                         float* logmel, float* feats, void* workspace, size_t workspace_bytes, void* stream, const LogmelOut& lmo) {
  MST_REQUIRE(p && stems4[0] && stems4[1] && stems4[2] && stems4[3], "mst_melfeat_forward: NULL plan/stems");
  MST_REQUIRE(clip_stride >= 2LL * T, "mst_melfeat_forward_stems: clip_stride %lld < 2*T", clip_stride);
  MST_REQUIRE(B > 0 && T > p->n_fft / 2, "mst_melfeat_forward: need B>0 and T > n_fft/2 (reflect pad); B=%d T=%d", B, T);
  MST_REQUIRE((long long)B * 8 * T < (1LL << 40), "mst_melfeat_forward: input too large");
  const size_t need = mst_melfeat_workspace_bytes(p, B, T);
  if (!workspace || workspace_bytes < need)
    return mst::fail(MST_ENOMEM, "mst_melfeat_forward: workspace %zu B < required %zu B", workspace_bytes, need);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int F = 1 + T / p->hop;
  KParams kp{};
  for (int i = 0; i < 4; ++i) kp.stem[i] = stems4[i];
  kp.clip_stride = clip_stride;
  kp.logmel = logmel, kp.partials = reinterpret_cast<float*>(workspace);
  kp.window = p->d_window, kp.tw = p->d_tw, kp.tw2 = p->d_tw2, kp.tw2_count = p->tw2_count, kp.post = p->d_post, kp.melw = p->d_melw, kp.lanebands = p->d_lanebands;
  kp.B = B, kp.T = T, kp.F = F, kp.M = p->n_mels, kp.hop = p->hop;
  kp.tw_count = p->tw_count, kp.nnz = p->nnz;
  for (int r = 0; r < 4; ++r) kp.glen[r] = p->glen[r], kp.goff[r] = p->goff[r];
  kp.frames_per_run = frames_per_run_of(p, B, F);
  kp.runs_per_clip = runs_per_clip(p, B, F);
  kp.pstride = pstride_of(p);
  bool base16 = true;
  for (int i = 0; i < 4; ++i) base16 = base16 && (reinterpret_cast<uintptr_t>(stems4[i]) & 15) == 0;
  kp.vec_ok = base16 && (T % 2 == 0) && (p->hop % 2 == 0) && (clip_stride % 2 == 0);
  kp.vec4_ok = base16 && (T % 4 == 0) && (p->hop % 4 == 0) && (clip_stride % 4 == 0);
  const int nc = p->nc;
  const int nf = nf_of(p->n_fft);
  const bool pair = p->n_fft <= 1024;
  size_t lds = (size_t)(nc + (pair ? p->tw2_count : p->tw_count + nc) + kWaves * nf * (nc + nc / 8)) * sizeof(float2) +
               (size_t)(((p->nnz + 3) & ~3) + 2 * 2 * p->n_mels * kTileStride) * sizeof(float);
  kp.tile_bufs = 2;
  if (lds > 160 * 1024) {  // fall back to a single output tile
    lds -= (size_t)2 * p->n_mels * kTileStride * sizeof(float);
    kp.tile_bufs = 1;
  }
  // the reduction buffers alias the FFT scratch: kWaves*4*NB*64 + kWaves*12 floats must fit
  const size_t red_need = (size_t)(kWaves * 4 * p->nb * 64 + kWaves * 12) * sizeof(float);
  MST_REQUIRE(red_need <= (size_t)kWaves * nf * (nc + nc / 8) * sizeof(float2), "internal: reduction buffer");
  MST_REQUIRE(lds <= 160 * 1024, "mst_melfeat_forward: LDS %zu B exceeds 160 KiB", lds);
  hipError_t e = hipErrorInvalidValue;
  // sliding-window kernel for the standard configuration (MST_STAGE_A=spw | generic selects the older kernels)
  const bool use_v2 = p->v2_ok && !stage_a_env_old();
  const bool use_v4 = p->v4_ok && !stage_a_env_old();
  MST_REQUIRE(lmo.layout == MST_LOGMEL_REF || use_v2 || use_v4, "mst_melfeat_forward: channel-minor log-mel needs the sliding-window kernels");
  if (lmo.absmax) {
    MST_REQUIRE(lmo.layout != MST_LOGMEL_REF, "mst_melfeat_forward: absmax comes with the channel-minor layouts only");
    MST_HIP_CHECK(hipMemsetAsync(lmo.absmax, 0, (size_t)B * sizeof(unsigned), st));
  }
  if (use_v4) {
    K4Params k4{};
    for (int i = 0; i < 4; ++i) k4.stem[i] = stems4[i];
    k4.clip_stride = clip_stride, k4.logmel = logmel, k4.partials = reinterpret_cast<float*>(workspace);
    k4.out_mode = lmo.layout, k4.logmel_lo = lmo.lo, k4.absmax = lmo.absmax;
    k4.window = p->d_window, k4.tw2 = p->d_v2_tw2, k4.tw3 = p->d_v2_tw3, k4.tw4 = p->d_v4_tw4, k4.segw = p->d_v4_segw;
    k4.pstart = p->d_v4_pstart, k4.pid = p->d_v4_pid, k4.bandtab = p->d_v4_bandtab;
    k4.B = B, k4.T = T, k4.F = F, k4.M = p->n_mels;
    k4.nslot = p->v4_nslot, k4.segw_count = p->v4_segw_count, k4.maxcnt = p->v4_maxcnt;
    for (int r = 0; r < 4; ++r) k4.glen[r] = p->v4_glen[r], k4.goff[r] = p->v4_goff[r];
    k4.fpw = v4_fpw(F), k4.runs_per_clip = v4_runs(F), k4.pstride = pstride_of(p);
    kp.runs_per_clip = k4.runs_per_clip;
    const size_t lds4 = (size_t)((mstpk::kTw2Rows + mstpk::kTw3Rows + 16) * 64 + p->v4_segw_count + 1024 + 8 * v4::kScr4) * sizeof(float2) +
                        (size_t)v4::kWPS * v4::kBLK * 4 * 128 * sizeof(float) + 16;
    MST_REQUIRE(lds4 <= 160 * 1024, "mst_melfeat_forward: LDS %zu B exceeds 160 KiB", lds4);
    const int grid4 = B * k4.runs_per_clip;
    e = pcm16 ? launch_melfeat_v2_2048<short>(k4, grid4, lds4, st) : launch_melfeat_v2_2048<float>(k4, grid4, lds4, st);
  } else if (use_v2) {
    K2Params k2{};
    for (int i = 0; i < 4; ++i) k2.stem[i] = stems4[i];
    k2.clip_stride = clip_stride, k2.logmel = logmel, k2.partials = reinterpret_cast<float*>(workspace);
    k2.out_mode = lmo.layout, k2.logmel_lo = lmo.lo, k2.absmax = lmo.absmax;
    k2.window = p->d_window, k2.tw2 = p->d_v2_tw2, k2.tw3 = p->d_v2_tw3, k2.segw = p->d_v2_segw;
    k2.segstart = p->d_v2_segstart, k2.segid = p->d_v2_segid, k2.bandtab = p->d_v2_bandtab, k2.maxcnt = p->v2_maxcnt;
    k2.B = B, k2.T = T, k2.F = F, k2.M = p->n_mels;
    k2.nslot = p->v2_nslot, k2.segw_count = p->v2_segw_count;
    for (int r = 0; r < kV2Slots; ++r) k2.glen[r] = p->v2_glen[r], k2.goff[r] = p->v2_goff[r];
    k2.fpw = v2_fpw(p, F), k2.runs_per_clip = v2_runs(p, F), k2.pstride = pstride_of(p);
    kp.runs_per_clip = k2.runs_per_clip;   // the finalise kernel walks the same records
    const int wps = p->v2_wps;
    const bool wide = p->n_mels > 128;   // 4 bands per lane, 4-frame blocks
    const size_t lds2 = (size_t)((mstpk::kTw2Rows + mstpk::kTw3Rows) * 64 + p->v2_segw_count + 4 * wps * mstpk::kScr) * sizeof(float2) +
                        (size_t)(wps * (wide ? 4 : (wps == 2 ? 8 : 4)) * 4 * (wide ? 256 : 128) + 1024) * sizeof(float) + 16;
    MST_REQUIRE(lds2 <= 160 * 1024, "mst_melfeat_forward: LDS %zu B exceeds 160 KiB", lds2);
    const int grid2 = B * k2.runs_per_clip;
    if (wide) e = pcm16 ? launch_melfeat_v2<short, 2, 4>(k2, grid2, lds2, st) : launch_melfeat_v2<float, 2, 4>(k2, grid2, lds2, st);
    else if (wps == 3) {
      const bool stdt = v2_table_is_standard(p) && !getenv("MST_V2_NOSTD");
      e = pcm16 ? launch_melfeat_v2<short, 3>(k2, grid2, lds2, st, stdt) : launch_melfeat_v2<float, 3>(k2, grid2, lds2, st, stdt);
    }
    else e = pcm16 ? launch_melfeat_v2<short, 2>(k2, grid2, lds2, st) : launch_melfeat_v2<float, 2>(k2, grid2, lds2, st);
  } else {
  const int grid = B * kp.runs_per_clip;
  // stem-per-wave-pair kernel for the standard configuration; the generic kernel covers everything else
  // stem-per-wave-pair kernel for the standard configuration; the generic kernel covers everything else
  const int wps = getenv("MST_SPW_WPS") ? atoi(getenv("MST_SPW_WPS")) : 3;
  auto spw_lds = [&](int w) {
    return (size_t)(nc + p->tw2_count + 4 * w * (p->n_fft + p->n_fft / 8)) * sizeof(float2) +
           (size_t)(((p->nnz + 3) & ~3) + (w == 2 ? 16 : 6) * (8 * p->n_mels + 1)) * sizeof(float);
  };
  const bool spw_ok = kp.vec_ok && kp.vec4_ok && p->hop * 4 == p->n_fft && p->n_mels <= 128 && p->nb == 2 &&
                      (p->n_fft == 512 || p->n_fft == 1024) && !getenv("MST_MELFEAT_GENERIC");
  const bool spw3 = spw_ok && wps == 3 && p->n_fft == 1024 && spw_lds(3) <= 160 * 1024;
  const bool spw = spw3 || (spw_ok && spw_lds(2) <= 160 * 1024);
  if (spw3) {
    e = pcm16 ? launch_melfeat_spw<1024, short, 3>(kp, grid, spw_lds(3), st) : launch_melfeat_spw<1024, float, 3>(kp, grid, spw_lds(3), st);
  } else if (spw) {
    const size_t lds_spw = spw_lds(2);
    if (p->n_fft == 1024)
      e = pcm16 ? launch_melfeat_spw<1024, short, 2>(kp, grid, lds_spw, st) : launch_melfeat_spw<1024, float, 2>(kp, grid, lds_spw, st);
    else
      e = pcm16 ? launch_melfeat_spw<512, short, 2>(kp, grid, lds_spw, st) : launch_melfeat_spw<512, float, 2>(kp, grid, lds_spw, st);
  } else {
#define MST_CASE(NF)                                                                                          \
  case NF:                                                                                                    \
    if (pcm16) e = (p->nb == 2) ? launch_melfeat<NF, 2, short>(kp, grid, lds, st) : launch_melfeat<NF, 4, short>(kp, grid, lds, st); \
    else e = (p->nb == 2) ? launch_melfeat<NF, 2, float>(kp, grid, lds, st) : launch_melfeat<NF, 4, float>(kp, grid, lds, st);       \
    break;
  switch (p->n_fft) {
    MST_CASE(512)
    MST_CASE(1024)
    MST_CASE(2048)
  }
#undef MST_CASE
  }
  }
  if (e != hipSuccess) return mst::fail(MST_EHIP, "melfeat_kernel launch failed: %s", hipGetErrorString(e));
  if (feats) {
    FParams fp{kp.partials, feats, p->n_mels, F, T, kp.runs_per_clip, kp.pstride, p->detailed_bins, p->feat_dim};
    const size_t flds = (size_t)(4 * p->n_mels + kNumScalars + 24 + 4 * (p->detailed_bins > 0 ? p->detailed_bins : 0)) *
                        sizeof(double);
    hipLaunchKernelGGL(melfeat_finalize_kernel, dim3(B), dim3(256), flds, st, fp);
    e = hipGetLastError();
    if (e != hipSuccess) return mst::fail(MST_EHIP, "melfeat_finalize_kernel launch failed: %s", hipGetErrorString(e));
  }
  return MST_OK;
}

}  // namespace
