// Augmentation chain on the device.  Replaces (reference barry-mir/mixing-style-transfer)
// AudioAugmenter.augment_stems src/mixing_utils.py:376-419 and apply_spectral_tilt :421-433, apply_compression
// :435-447, apply_bandwidth_limit :449-456, apply_reverb :458-479.  All random decisions (coins, gain, cutoff,
// impulse response) are drawn by the host in the reference's order and arrive in `mst_aug_clip` / `reverb_ir`.
//
//   aug_chain_kernel   x*gain -> biquad (tilt) -> compressor -> 1-2 biquads (low-pass): one workgroup per (channel, segment) of
//                      16 384 samples resident in LDS, float64 block-parallel IIR inside a segment, decoupled look-back between a
//                      stream's segments (see the kernel's comment);
//                      one read and one write of every channel that has a decision.  scipy.signal.sosfilt is a sequential
//                      fp64 DF2T recurrence; rounded to fp32 exactly where the reference calls `.float()`.
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

constexpr int kBlk = 512;     // overlap-save block (FFT size 1024)
constexpr int kNfft = 1024;

struct ChainCarry {           // what segment k of a stream publishes: its AGGREGATES Z (the filter states after its last sample when started
                              // from rest), from which every later segment folds its own start state (chain_scan).  Every word
  unsigned long long t[2], b[4], pad_[2];   // starts as kCarryEmpty and is written ONCE, with one 64-bit agent-scope atomic store
};
constexpr unsigned long long kCarryEmpty = ~0ull;   // (all ones: a NaN pattern no arithmetic produces; hipMemset 0xFF)
struct ChainParams {
  float* stems;              // [B][8][T]: written (and, in place, read)
  const float* src;          // read from: == stems (in place), or the tensor the reference would have cloned (mst_aug_apply_from)
  long long src_stride;      // floats between the clips of src
  const mst_aug_clip* dec;   // device copy [B]
  int T;
  long long clip_stride;     // floats between clips (8 * T when packed)
  int B, nseg;
  int* act;                  // [1 + B * 8]: number of streams with a decision, then their indices (aug_active_kernel)
  double* pow;               // [B * 8][2 filters][kFPow][16]: powers M^(2^k) of the chunk transitions (aug_powers_kernel)
  int* ticket;               // [2]: work-item counter, error word (zeroed per launch)
  ChainCarry* carry;         // [B * 8][nseg] (zeroed per launch)
};

// mixing_utils.py:435-447:  dB = 20 log10(|x| + 1e-8);  above the threshold: dB' = thr + (dB - thr) / ratio;  y = sign(x) 10^(dB' / 20).
// In closed form, with a = |x| + 1e-8 and t = 10^(thr / 20):  a <= t -> y = sign * a (the 1e-8 offset survives, as in the
// reference);  a > t -> y = sign * 10^(thr (1 - 1/ratio) / 20) * a^(1 / ratio).
//   mode 1  the reference's default setting (-20 dB, 4:1): 10^(-3/4) * a^(1/4) as two square roots instead of log10f + powf -- ~10
//           instructions instead of ~100, and closer to the real-valued result than the reference's own float32 log / pow chain
//           (which it matches to a few 1e-7 relative; the goldens hold at 1e-5);
//   mode 2  any threshold / ratio (mst_aug_stem.comp_threshold_db / comp_ratio): the constants come from the host in double.
struct Comp {
  int mode;
  float lin_thr, c, inv_ratio;
};
__device__ __forceinline__ float compress_f32(float x, const Comp& k) {
  const float a = fabsf(x) + 1e-8f;
  float y;
  if (k.mode == 1) y = a > 0.1f ? 0.17782794100389228f * sqrtf(sqrtf(a)) : a;
  else y = a > k.lin_thr ? k.c * powf(a, k.inv_ratio) : a;
  return x > 0.f ? y : (x < 0.f ? -y : 0.f);
}
__device__ __forceinline__ Comp comp_of(const mst_aug_stem& d) {
  Comp k{d.compress, 0.1f, 0.17782794100389228f, 0.25f};
  if (d.compress == 2) {
    const double thr = (double)d.comp_threshold_db, ir = 1.0 / (double)d.comp_ratio;
    k.lin_thr = (float)pow(10.0, thr / 20.0), k.c = (float)pow(10.0, thr * (1.0 - ir) / 20.0), k.inv_ratio = (float)ir;
  }
  return k;
}

// ---- the IIR chain  x * gain -> [tilt biquad] -> [compressor] -> [1-2 low-pass biquads]  (mixing_utils.py:389-456) in ONE kernel.
// scipy.signal.sosfilt is a sequential float64 DF2T recurrence over the whole channel.  Here a channel ("stream") is cut into
// segments of 16 384 samples and ONE WORKGROUP owns one segment, which lives in LDS: it is read from HBM once (whole 128-byte
// lines), every pass over it reads and writes LDS, and it is written back once -- the chain moves its algorithmic bytes, one
// read and one write of every stream that has a decision (the three-launch form of round 3 read a filtered stream three times
// and wrote it twice).  Inside a segment the recurrences are block-parallel, exactly as before: a thread owns a chunk of 32
// consecutive samples;
//   (1) zero-state pass: the chunk's final filter state when started from rest (z_c);
//   (2) scan: start state of every chunk, s_c = M^c s_seg + sum_{j<c} M^(c-1-j) z_j with M = A^32 (the homogeneous transition
//       of a chunk; the powers M^(2^k) come from aug_powers_kernel, once per stream), as a doubling prefix inside each wave, a
//       serial carry over the 8 waves (M^64) and a lane-dependent power M^lane of the wave's carry; s_seg, the state at the
//       segment's start, comes from the EARLIER SEGMENTS' aggregates (decoupled look-back across workgroups: chain_scan);
//   (3) response pass: every chunk re-run from its true start state -- the reference's arithmetic sample for sample (float64
//       DF2T, sosfilt's section order, `.float()` after each filter); only the chunk start states are obtained differently.
// The tilt response pass also applies gain and compressor and runs the low-pass zero-state pass on the values it has just
// produced.  66 KB of LDS per workgroup: two workgroups per CU cover each other's loads, barriers and look-back polls
// (measured: 512 threads x 32 samples 0.47 ms with every effect on every one of 192 streams; x 64 samples 0.56; 256 x 32 0.45).
#ifndef MST_AUG_FT
#define MST_AUG_FT 512
#define MST_AUG_FL 32
#define MST_AUG_FB 16
#endif
constexpr int kFT = MST_AUG_FT, kFL = MST_AUG_FL, kFS = kFT * kFL, kFP = kFL + 1;   // threads, samples per chunk, samples per segment, LDS row pitch
constexpr int kFW = kFT / 64, kFB = MST_AUG_FB;
constexpr int kFPow = (kFT == 1024 ? 10 : kFT == 512 ? 9 : 8) + 1;    // powers M^(2^k) kept: the last one spans a whole segment
static_assert((1 << (kFPow - 1)) == kFT, "kFT must be 256, 512 or 1024");                               // waves; samples per register batch of the serial walks

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

struct ChainLds {
  float tile[kFT * kFP];
  double Mp[2][kFPow][16]; // [filter: 0 tilt, 1 low-pass][k][4 x 4]: M^(2^k), k = 0 .. log2(kFT), zero outside the D x D block
  double wtot[kFW][4];     // inclusive prefix of every wave's last lane
  double seg[4];           // state at the start of this segment
  double zin[kFT];         // the earlier segments' aggregates, [segment][D]
  int item;                // this workgroup's ticket
};

// powers M^(2^k), k = 0 .. kFPow - 1, of the chunk transition of a cascade of NS biquads (D = 2 NS states) into Mp[.][16]
__device__ __forceinline__ void chain_powers(double (*Mp)[16], const double* sos, int NS, int tid) {
  const int D = 2 * NS;
  if (tid < 16) Mp[0][tid] = 0.0;
  __syncthreads();
  if (tid < D) {   // column tid of M = A^kFL: homogeneous response to a unit initial state
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < 4; ++i) s[i] = (i == tid) ? 1.0 : 0.0;
    const Coef k0 = coef_of(sos), k1 = coef_of(sos + (NS > 1 ? 6 : 0));
    for (int n = 0; n < kFL; ++n) {
      double v = biquad_step(k0, s[0], s[1], 0.0);
      if (NS > 1) v = biquad_step(k1, s[2], s[3], v);
    }
    for (int i = 0; i < D; ++i) Mp[0][i * 4 + tid] = s[i];
  }
  __syncthreads();
  for (int k = 1; k < kFPow; ++k) {
    if (tid < 16) {
      const int i = tid >> 2, j = tid & 3;
      double t = 0.0;
      for (int q = 0; q < 4; ++q) t += Mp[k - 1][i * 4 + q] * Mp[k - 1][q * 4 + j];
      Mp[k][tid] = t;
    }
    __syncthreads();
  }
}

// one wave per stream: the powers of both filters' chunk transitions, ONCE per stream (every segment workgroup of the stream
// reads them: 2.8 KB) instead of once per segment workgroup (64 dependent recurrence steps + 10 squarings behind barriers)
__global__ __launch_bounds__(64) void aug_powers_kernel(const mst_aug_clip* dec, double* pow) {
  __shared__ double Mp[kFPow][16];
  const int stream = blockIdx.x, tid = threadIdx.x;
  const mst_aug_stem& d = dec[stream >> 3].stem[(stream & 7) >> 1];
  for (int f = 0; f < 2; ++f) {
    const int NS = f == 0 ? (d.tilt != 0 ? 1 : 0) : d.bw_sections;
    if (NS == 0) continue;   // block-uniform
    chain_powers(Mp, f == 0 ? d.tilt_sos : d.bw_sos, NS, tid);
    for (int i = tid; i < kFPow * 16; i += 64) pow[((size_t)stream * 2 + f) * (kFPow * 16) + i] = Mp[i >> 4][i & 15];
    __syncthreads();
  }
}

template <int D>
__device__ __forceinline__ void matvec(const double* M, const double (&v)[D], double (&out)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) t += M[i * 4 + j] * v[j];
    out[i] = t;
  }
}

// Hand-over between the segments of one stream (different workgroups, possibly on different XCDs, whose L2s are not coherent): the
// state words travel as 64-bit agent-scope ATOMIC stores / loads (relaxed: they go past the non-coherent cache levels; no release /
// acquire fence -- on this chip a release writes back the XCD's whole L2, 128 KB of freshly stored samples per workgroup, and made
// every hop ~15 us).  No flag and no ordering between the words is needed: each word starts as kCarryEmpty and is valid the
// moment it reads as anything else.  Work items are handed out by an atomic ticket in the order the workgroups START, and the
// segments of a stream hold consecutive tickets in their own order: whoever is waited for started earlier and is running or
// done -- no dispatch order is assumed.  The poll is bounded (~1 s; then the error word is raised and the state is NaN, which
// the segment's samples inherit: visible, never silent), so the grid always drains.
__device__ __forceinline__ double chain_wait(const unsigned long long* word, int* err) {
  for (int it = 0; it < (1 << 22); ++it) {
    const unsigned long long v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v != kCarryEmpty) return __longlong_as_double((long long)v);
    __builtin_amdgcn_s_sleep(2);
  }
  atomicExch(err, 1);
  return __builtin_nan("");   // never silent: a segment whose predecessor did not deliver turns into NaN samples
}
__device__ __forceinline__ void chain_post(unsigned long long* word, double v) {
  __hip_atomic_store(word, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// z: in = zero-state final state of this thread's chunk, out = the chunk's true start state.
// Across segments (different workgroups) this is a DECOUPLED look-back: thread 0 folds the waves' totals into the segment's
// aggregate Z (its final state from a zero start: needs nothing from other segments) and publishes it (`mine`); then the
// workgroup reads the aggregates of ALL earlier segments of the stream (`first`, records `stride` words apart) -- in parallel,
// one word per thread, polling the few that are not there yet -- and thread 0 folds them, s <- M^kFT s + Z_j for j = 0 .. nprev - 1,
// into the state at this segment's start.  No segment waits for another one's RESULT, only for its aggregate, and all
// aggregates of a stream are produced at about the same time: the serial hand-over chain of the first version (14 hops per
// stream and filter) is one hop deep.
template <int D>
__device__ __forceinline__ void chain_scan(double (&z)[D], const double (*Mp)[16], double (*wtot)[4], double* seg, double* zin, int tid,
                                           const unsigned long long* first, int stride, int nprev, unsigned long long* mine, int* err) {
  const int lane = tid & 63, wave = tid >> 6;
  double P[D];
#pragma unroll
  for (int i = 0; i < D; ++i) P[i] = z[i];
#pragma unroll
  for (int k = 0; k < 6; ++k) {   // inclusive prefix inside the wave: P_lane = sum_{j <= lane} M^(lane - j) z_j
    double v[D], t[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      v[i] = __shfl_up(P[i], 1u << k, 64);
      if (lane < (1 << k)) v[i] = 0.0;
    }
    matvec<D>(Mp[k], v, t);
#pragma unroll
    for (int i = 0; i < D; ++i) P[i] += t[i];
  }
  if (lane == 63)
#pragma unroll
    for (int i = 0; i < D; ++i) wtot[wave][i] = P[i];
  __syncthreads();
  if (tid == 0) {
    double Z[D], t[D];
#pragma unroll
    for (int i = 0; i < D; ++i) Z[i] = 0.0;
    for (int w = 0; w < kFW; ++w) {   // Z <- M^64 Z + W_w
      matvec<D>(Mp[6], Z, t);
#pragma unroll
      for (int i = 0; i < D; ++i) Z[i] = t[i] + wtot[w][i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) chain_post(mine + i, Z[i]);
  }
  for (int i = tid; i < nprev * D; i += kFT) zin[i] = chain_wait(first + (size_t)(i / D) * stride + (i % D), err);
  __syncthreads();
  if (tid == 0) {
    double s[D], t[D];
#pragma unroll
    for (int i = 0; i < D; ++i) s[i] = 0.0;
    for (int j = 0; j < nprev; ++j) {
      matvec<D>(Mp[kFPow - 1], s, t);
#pragma unroll
      for (int i = 0; i < D; ++i) s[i] = t[i] + zin[j * D + i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) seg[i] = s[i];
  }
  __syncthreads();
  double C[D];   // state at the start of this wave's first chunk: C_0 = seg, C_(w+1) = M^64 C_w + W_w
#pragma unroll
  for (int i = 0; i < D; ++i) C[i] = seg[i];
  for (int w = 0; w < wave; ++w) {
    double t[D];
    matvec<D>(Mp[6], C, t);
#pragma unroll
    for (int i = 0; i < D; ++i) C[i] = t[i] + wtot[w][i];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {   // M^lane C
    double t[D];
    matvec<D>(Mp[k], C, t);
    const bool bit = (lane >> k) & 1;
#pragma unroll
    for (int i = 0; i < D; ++i) C[i] = bit ? t[i] : C[i];
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const double e = __shfl_up(P[i], 1u, 64);   // exclusive prefix
    z[i] = C[i] + (lane == 0 ? 0.0 : e);
  }
  __syncthreads();   // every thread has read seg and wtot (the next scan overwrites them)
}

// one wave: the streams that have a decision -> act[0] = their number, act[1 ..] = their indices (clip * 8 + channel)
// all != 0 (out of place): every stream -- one without a decision is copied by the chain kernel
__global__ __launch_bounds__(64) void aug_active_kernel(const mst_aug_clip* dec, int B, int* act, int all) {
  const int lane = threadIdx.x;
  int n = 0;
  for (int s0 = 0; s0 < B * 8; s0 += 64) {
    const int stream = s0 + lane;
    bool on = false;
    if (stream < B * 8) {
      const mst_aug_stem& d = dec[stream >> 3].stem[(stream & 7) >> 1];
      on = all || d.gain != 1.0f || d.tilt != 0 || d.compress != 0 || d.bw_sections > 0;
    }
    const unsigned long long m = __ballot(on);
    if (on) act[1 + n + __popcll(m & ((1ull << lane) - 1ull))] = stream;
    n += __popcll(m);
  }
  if (lane == 0) act[0] = n;
}

// grid (B * 8 * nseg), kFT threads: one workgroup per (stream, segment), handed out by ticket
__global__ __launch_bounds__(kFT) void aug_chain_kernel(const ChainParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  ChainLds& lds = *reinterpret_cast<ChainLds*>(chain_smem);
  const int tid = threadIdx.x;
  if (tid == 0) lds.item = atomicAdd(p.ticket, 1);
  __syncthreads();
  const int item = lds.item;
  if (item >= p.act[0] * p.nseg) return;   // block-uniform: more workgroups than (streams with a decision) x segments
  const int a = item / p.nseg, sg = item - a * p.nseg, stream = p.act[1 + a];
  const mst_aug_stem& d = p.dec[stream >> 3].stem[(stream & 7) >> 1];
  const bool has_t = d.tilt != 0, has_c = d.compress != 0, has_b = d.bw_sections > 0;
  float* x = p.stems + (size_t)(stream >> 3) * p.clip_stride + (size_t)(stream & 7) * p.T;
  const float* xin = p.src + (size_t)(stream >> 3) * p.src_stride + (size_t)(stream & 7) * p.T;
  const int T = p.T;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(xin)) & 15) == 0;   // 16-byte accesses (else four 4-byte ones)
  const Coef kt = coef_of(d.tilt_sos), kb0 = coef_of(d.bw_sos), kb1 = coef_of(d.bw_sos + 6);
  const Comp kc = comp_of(d);
  const float gain = d.gain;
  const bool two = d.bw_sections > 1;
  ChainCarry* const mine = p.carry + (size_t)a * p.nseg + sg;
  const ChainCarry* const first = mine - sg;   // the stream's first segment (records 8 words apart)
  // mover mapping: piece i of a segment = 16-byte unit i * kFT + tid -> samples 4 (i kFT + tid) .. + 3 of the segment
  const int n0 = sg * kFS;
#pragma unroll
  for (int i = 0; i < kFL / 4; ++i) {
    const int u = 4 * (i * kFT + tid);
    const long long n = (long long)n0 + u;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n + 3 < T) {
      if (vec) v = *reinterpret_cast<const float4*>(xin + n);
      else v = make_float4(xin[n], xin[n + 1], xin[n + 2], xin[n + 3]);
    } else if (n < T) {
      v.x = xin[n];
      if (n + 1 < T) v.y = xin[n + 1];
      if (n + 2 < T) v.z = xin[n + 2];
    }
    float* q = lds.tile + (u / kFL) * kFP + (u % kFL);
    q[0] = v.x, q[1] = v.y, q[2] = v.z, q[3] = v.w;
  }
  for (int i = tid; i < 2 * kFPow * 16; i += kFT)   // both filters' powers (zeros where a filter is off: never used)
    (&lds.Mp[0][0][0])[i] = ((i < kFPow * 16) ? has_t : has_b) ? p.pow[(size_t)stream * 2 * (kFPow * 16) + i] : 0.0;
  __syncthreads();
  float* row = lds.tile + tid * kFP;
  const int nv = max(0, min(kFL, T - (n0 + tid * kFL)));   // samples of this thread's chunk inside the clip
  double ts[2] = {0.0, 0.0}, bs[4] = {0.0, 0.0, 0.0, 0.0};
  // serial walks over the chunk, kFB samples at a time through registers (one LDS round trip per batch, not per sample)
  auto walk = [&](auto&& f, bool store) __attribute__((always_inline)) {
    if (nv == kFL) {
#pragma unroll 1
      for (int j0 = 0; j0 < kFL; j0 += kFB) {
        float v[kFB];
#pragma unroll
        for (int q = 0; q < kFB; ++q) v[q] = row[j0 + q];
#pragma unroll
        for (int q = 0; q < kFB; ++q) v[q] = f(v[q]);
        if (store) {
#pragma unroll
          for (int q = 0; q < kFB; ++q) row[j0 + q] = v[q];
        }
      }
    } else {
      for (int j = 0; j < nv; ++j) {   // the clip's last, ragged chunk
        const float o = f(row[j]);
        if (store) row[j] = o;
      }
    }
  };
  if (has_t) {   // (1) + (2) of the tilt biquad
    walk([&](float v) { (void)biquad_step(kt, ts[0], ts[1], (double)(v * gain)); return v; }, false);
    chain_scan<2>(ts, lds.Mp[0], lds.wtot, lds.seg, lds.zin, tid, first->t, 8, sg, mine->t, p.ticket + 1);
  }
  // (3) of the tilt biquad, gain, compressor; (1) of the low-pass
  walk([&](float v) {
    float y = v * gain;
    if (has_t) y = (float)biquad_step(kt, ts[0], ts[1], (double)y);
    if (has_c) y = compress_f32(y, kc);
    if (has_b) {
      const double w = biquad_step(kb0, bs[0], bs[1], (double)y);
      if (two) (void)biquad_step(kb1, bs[2], bs[3], w);
    }
    return y;
  }, true);
  if (has_b) {
    if (two) {
      chain_scan<4>(bs, lds.Mp[1], lds.wtot, lds.seg, lds.zin, tid, first->b, 8, sg, mine->b, p.ticket + 1);
    } else {
      double b2[2] = {bs[0], bs[1]};
      chain_scan<2>(b2, lds.Mp[1], lds.wtot, lds.seg, lds.zin, tid, first->b, 8, sg, mine->b, p.ticket + 1);
      bs[0] = b2[0], bs[1] = b2[1];
    }
    walk([&](float v) {
      double w = biquad_step(kb0, bs[0], bs[1], (double)v);
      if (two) w = biquad_step(kb1, bs[2], bs[3], w);
      return (float)w;
    }, true);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kFL / 4; ++i) {
    const int u = 4 * (i * kFT + tid);
    const long long n = (long long)n0 + u;
    const float* q = lds.tile + (u / kFL) * kFP + (u % kFL);
    if (n + 3 < T) {
      if (vec) *reinterpret_cast<float4*>(x + n) = make_float4(q[0], q[1], q[2], q[3]);
      else x[n] = q[0], x[n + 1] = q[1], x[n + 2] = q[2], x[n + 3] = q[3];
    } else {
      for (int e = 0; e < 4; ++e)
        if (n + e < T) x[n + e] = q[e];
    }
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
  // (consecutive windows on one XCD: a window shares half of its samples with each neighbour, so the second read of every sample
  //  hits that XCD's L2 -- blocks x and x + 8 of a clip share an XCD, mst::xcd_remap hands each XCD a contiguous run)
  const int b = blockIdx.y, item = mst::xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
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
  // spectrum layout [r / 2][lane][r & 1]: a lane's registers 2 k, 2 k + 1 are one 16-byte unit -- rev_mac_ifft_kernel reads them with
  // ds_read_b128 (256 B/clk/CU; the ds_read2st64_b64 pairs hipcc makes of a [r][lane] layout move half of that)
  float4* dst = reinterpret_cast<float4*>(KIND == 0 ? p.G + ((size_t)b * p.NP + item) * 1024 : p.X + ((size_t)b * p.NX + item) * 1024) + lane;
#pragma unroll
  for (int k = 0; k < 8; ++k) dst[k * 64] = make_float4(v[2 * k].x, v[2 * k].y, v[2 * k + 1].x, v[2 * k + 1].y);
}

// Spectral multiply-accumulate over the NP partitions + inverse FFT + redistribution.  A workgroup = 8 waves = 8 consecutive
// output blocks j; wave w needs X[j0 + w - q] and G[q] at step q.  Every spectrum is 8 KB: read per wave from global memory
// that was 2 x 44 x 8 KB per output block, 7 GB through L2 per step (the kernel ran at the L2 rate).  Here the workgroup
// shares them through LDS: G[q] is loaded once per step for all 8 waves (double-buffered), and the X windows slide -- step q
// needs ONE new block, X[j0 - q], which replaces the block only step q - 1's last wave still used (a ring of 9 slots, so
// that the write of step q never touches what step q - 1 reads: one barrier per step).  Global traffic per step: 16 KB per
// workgroup instead of 128 KB.
constexpr int kRevWaves = 8, kRevRing = kRevWaves + 3;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// acc + a * w (complex) as two packed FMAs, in the nesting the scalar form had: (acc + (-a.y w.y, a.y w.x)) + (a.x w.x, a.x w.y).
// Inline asm: hipcc builds the swapped / negated operands of the packed form with v_mov + v_xor (2 moves per FMA).
__device__ __forceinline__ v2f cmac(v2f acc, v2f a, v2f w) {
  v2f t, d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(t) : "v"(a), "v"(w), "v"(acc));   // (-a.y w.y, a.y w.x) + acc
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(w), "v"(t));                                    // (a.x w.x, a.x w.y) + t
  return d;
}
struct RevLds {
  float2 tw[FftPlan<kNfft>::TW];
  union {
    struct {
      float2 X[kRevRing][1024];
      float2 G[4][1024];
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
  // Consecutive workgroups of a clip on ONE XCD: workgroup g needs the windows X[8 g - 43 .. 8 g + 7] and all 44 G spectra -- its
  // neighbour 43 of the same 51 windows and the same G.  Dealt round-robin (the hardware's order), a clip's workgroups on one XCD
  // are 8 apart and share nothing: every spectrum came from HBM once per workgroup (0.88 GB per step for 0.1 GB of spectra).
  const int wg = mst::xcd_remap(blockIdx.x, gridDim.x);
  const int jb = p.j0 + wg * kRevWaves;   // first output block of this workgroup
  const int j = jb + wave;
  const bool mine = wg * kRevWaves + wave < p.nj;
  const float4* Gg = reinterpret_cast<const float4*>(p.G + (size_t)b * p.NP * 1024);
  const float4* Xg = reinterpret_cast<const float4*>(p.X + (size_t)b * p.NX * 1024);
  auto slot = [](int i) { return ((i % kRevRing) + kRevRing) % kRevRing; };
  // prologue: the 8 windows of step 0
  for (int w = 0; w < kRevWaves; ++w) {
    const int i = jb + w;
    const float4 v = (i >= 0 && i < p.NX) ? Xg[(size_t)i * 512 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(lds.u.r.X[slot(i)])[tid] = v;
  }
  // The two 8 KB blocks a step needs (G[q] and the one new window X[jb - q]) are fetched kRevAhead steps ahead into registers: a
  // step's arithmetic is ~0.3 us, a global load ~1.5 us -- one step of look-ahead left the loop waiting on memory 44 times per
  // workgroup (the kernel ran 0.36 ms for 0.05 ms of multiply-adds).
  constexpr int kRevAhead = 4;
  float4 gq[kRevAhead], xq[kRevAhead];
  auto issue = [&](int q, float4& gdst, float4& xdst) __attribute__((always_inline)) {
    if (q < p.NP) {   // block-uniform
      gdst = Gg[(size_t)q * 512 + tid];
      const int i = jb - q;
      xdst = (q > 0 && i >= 0 && i < p.NX) ? Xg[(size_t)i * 512 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);   // (step 0's window is in place)
    }
  };
#pragma unroll
  for (int u = 0; u < kRevAhead; ++u) gq[u] = xq[u] = make_float4(0.f, 0.f, 0.f, 0.f), issue(u, gq[u], xq[u]);
  v2f acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = v2f{0.f, 0.f};
  // TWO steps per barrier: the pair (q, q + 1) needs the windows jb - q - 1 .. jb - q + 7 and two G blocks; what it writes (two new
  // windows, two G blocks) goes to slots the previous pair -- which other waves may still be reading -- does not use: a ring of
  // kRevWaves + 3 windows and four G buffers.
  for (int q0 = 0; q0 < p.NP; q0 += kRevAhead) {
#pragma unroll
    for (int h = 0; h < kRevAhead / 2; ++h) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int u = 2 * h + e, q = q0 + u;
        if (q < p.NP) {   // block-uniform
          reinterpret_cast<float4*>(lds.u.r.G[q & 3])[tid] = gq[u];
          if (q > 0) reinterpret_cast<float4*>(lds.u.r.X[slot(jb - q)])[tid] = xq[u];
        }
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 2; ++e) issue(q0 + 2 * h + e + kRevAhead, gq[2 * h + e], xq[2 * h + e]);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int q = q0 + 2 * h + e, i = j - q;
        if (q < p.NP && mine && i >= 0 && i < p.NX) {   // wave-uniform
          const f32x4* x = reinterpret_cast<const f32x4*>(lds.u.r.X[slot(i)]) + lane;
          const f32x4* g = reinterpret_cast<const f32x4*>(lds.u.r.G[q & 3]) + lane;
          f32x4 a[8], w[8];   // all 16 reads (16 bytes each) in flight before the first multiply-add: one LDS round trip per step
#pragma unroll
          for (int k = 0; k < 8; ++k) a[k] = x[k * 64], w[k] = g[k * 64];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            acc[2 * k] = cmac(acc[2 * k], v2f{a[k].x, a[k].y}, v2f{w[k].x, w[k].y});
            acc[2 * k + 1] = cmac(acc[2 * k + 1], v2f{a[k].z, a[k].w}, v2f{w[k].z, w[k].w});
          }
        }
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
    const v2f y = acc[out_reg<kNfft>(in_q<kNfft>(r))];
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
  size_t dec, act, pow, lb, prop, epart, G, X, total;   // lb: ticket words + carry records (zeroed per launch)
  size_t lb_bytes;
  int nseg, NP, NX, D, j0, nj;
};

AugLayout aug_layout(int B, int T, int L) {
  AugLayout a{};
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
  a.nseg = (T + kFS - 1) / kFS;
  a.act = take((size_t)(1 + B * 8) * sizeof(int));
  a.pow = take((size_t)B * 8 * 2 * kFPow * 16 * sizeof(double));
  a.lb_bytes = 64 + (size_t)B * 8 * a.nseg * sizeof(ChainCarry);
  a.lb = take(a.lb_bytes);
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
  return mst_aug_apply_from(decisions, B, T, stems_inout, clip_stride, stems_inout, clip_stride, reverb_ir, ir_len, workspace,
                            workspace_bytes, stream);
}

int mst_aug_apply_from(const mst_aug_clip* decisions, int B, int T, const float* src, long long src_stride, float* stems_inout,
                       long long clip_stride, const float* reverb_ir, int ir_len, void* workspace, size_t workspace_bytes,
                       void* stream) {
  MST_REQUIRE(decisions && stems_inout && src, "mst_aug_apply: NULL argument");
  MST_REQUIRE(clip_stride >= (long long)8 * T, "mst_aug_apply_strided: clip_stride %lld < 8 * T", clip_stride);
  MST_REQUIRE(src_stride >= (long long)8 * T, "mst_aug_apply_from: src_stride %lld < 8 * T", src_stride);
  const bool in_place = src == stems_inout;
  MST_REQUIRE(!in_place || src_stride == clip_stride, "mst_aug_apply_from: src == dst with different strides");
  if (!in_place && B > 0 && T > 0) {   // the two tensors must not overlap: a stream's segments are read and written by different workgroups
    const char* a0 = reinterpret_cast<const char*>(src);
    const char* a1 = a0 + ((size_t)(B - 1) * src_stride + (size_t)8 * T) * sizeof(float);
    const char* b0 = reinterpret_cast<const char*>(stems_inout);
    const char* b1 = b0 + ((size_t)(B - 1) * clip_stride + (size_t)8 * T) * sizeof(float);
    MST_REQUIRE(a1 <= b0 || b1 <= a0, "mst_aug_apply_from: src and dst overlap");
  }
  MST_REQUIRE(B > 0 && T > 0 && ir_len >= 0, "mst_aug_apply: bad sizes B=%d T=%d ir_len=%d", B, T, ir_len);
  bool any_rev = false;
  for (int b = 0; b < B; ++b) {
    any_rev = any_rev || decisions[b].reverb != 0;
    MST_REQUIRE(decisions[b].reverb >= 0 && decisions[b].reverb <= 2, "mst_aug_apply: bad reverb flag");
    for (int s = 0; s < 4; ++s)
    {
      MST_REQUIRE(decisions[b].stem[s].bw_sections >= 0 && decisions[b].stem[s].bw_sections <= 2,
                  "mst_aug_apply: bw_sections must be 0..2");
      MST_REQUIRE(decisions[b].stem[s].compress >= 0 && decisions[b].stem[s].compress <= 2, "mst_aug_apply: compress must be 0, 1 or 2");
      MST_REQUIRE(decisions[b].stem[s].compress != 2 || (decisions[b].stem[s].comp_ratio > 0.f && decisions[b].stem[s].comp_threshold_db < 160.f &&
                                                         decisions[b].stem[s].comp_threshold_db > -160.f),
                  "mst_aug_apply: compress = 2 needs comp_ratio > 0 and a finite comp_threshold_db");
    }
  }
  MST_REQUIRE(!any_rev || (reverb_ir && ir_len > 0), "mst_aug_apply: reverb requested but no impulse response");
  const AugLayout L = aug_layout(B, T, any_rev ? ir_len : 0);
  if (!workspace || workspace_bytes < L.total)
    return mst::fail(MST_ENOMEM, "mst_aug_apply: workspace %zu B < required %zu B", workspace_bytes, L.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  mst_aug_clip* ddec = reinterpret_cast<mst_aug_clip*>(ws + L.dec);
  MST_HIP_CHECK(hipMemcpyAsync(ddec, decisions, (size_t)B * sizeof(mst_aug_clip), hipMemcpyHostToDevice, st));
  bool any_chain = false;
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < 4; ++s) {
      const mst_aug_stem& d = decisions[b].stem[s];
      any_chain = any_chain || d.gain != 1.0f || d.tilt != 0 || d.compress != 0 || d.bw_sections > 0;
    }
  any_chain = any_chain || !in_place;   // out of place the chain is also the copy
  if (any_chain) {
    int* lb = reinterpret_cast<int*>(ws + L.lb);
    ChainParams cp{stems_inout, src, src_stride, ddec, T, clip_stride, B, L.nseg, reinterpret_cast<int*>(ws + L.act), reinterpret_cast<double*>(ws + L.pow), lb,
                   reinterpret_cast<ChainCarry*>(ws + L.lb + 64)};
    static_assert(sizeof(ChainCarry) == 64, "one carry record per 64 bytes");
    MST_HIP_CHECK(hipMemsetAsync(lb, 0, 64, st));
    MST_HIP_CHECK(hipMemsetAsync(ws + L.lb + 64, 0xFF, L.lb_bytes - 64, st));   // every carry word = kCarryEmpty
    hipLaunchKernelGGL(aug_active_kernel, dim3(1), dim3(64), 0, st, ddec, B, cp.act, in_place ? 0 : 1);
    hipLaunchKernelGGL(aug_powers_kernel, dim3(B * 8), dim3(64), 0, st, ddec, cp.pow);
    static unsigned long long chain_attr = 0;   // per-device bit mask: the dynamic-LDS limit belongs to the device
    if (mst::first_use_on_device(chain_attr))
      MST_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(aug_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)sizeof(ChainLds)));
    hipLaunchKernelGGL(aug_chain_kernel, dim3(B * 8 * L.nseg), dim3(kFT), sizeof(ChainLds), st, cp);
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
