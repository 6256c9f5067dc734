// Stage B: FiLM MLP -> per-band [Conv7x7 + BN(eval) + FiLM + ReLU + MaxPool] x2 -> attention pooling -> projection.
//
// Replaces (reference barry-mir/mixing-style-transfer, eval mode): MixingFeatureEncoder.forward
// src/model.py:410-464, SubSpectrogramCNN.forward :127-157 (looped over sub-bands :345-362), the concat/view
// :332-367 and AttentionPooling.forward :187-211.
//
// Convolutions are implicit GEMMs on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32, a k-ordered fmaf
// chain): rows = output positions, columns = output channels, K = (tap, input channel).  The row order inside
// a wave tile is chosen so that every max-pool window lies in the accumulator registers of ONE lane: BN(eval),
// FiLM, ReLU and the max-pool run in registers and only pooled activations reach HBM.
//   * one workgroup = 8 waves (2 per SIMD), persistent over "sets" of 8 wave tiles of one sub-band;
//   * weights stream through LDS in chunks of 4 input channels x 49 taps, pre-swizzled on the host into MFMA
//     B-fragment order (one conflict-free ds_read_b32 per fragment);
//   * each wave owns a private LDS patch (4 channels x (rows+6) x (cols+6)) of its tile: no inter-wave
//     hand-off except the shared weight chunk (one barrier per chunk);
//   * the next chunk (weights + patch) is prefetched into registers while the current one is on the MFMAs.
#include "common.h"

#include <cmath>
#include <type_traits>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kConvThreads = 512;
constexpr int kConvWaves = 8;

// ------------------------------------------------------------------------------------------
// FiLM MLP + folded per-(clip, band, channel) affine:  y = A * conv + C  with
//   A = gamma * s,  C = gamma * t + beta,  s = bn_w / sqrt(var + eps),  t = s * (conv_b - mean) + bn_b
// ------------------------------------------------------------------------------------------
struct FilmParams {
  const float* feats;
  const float *w0t, *b0, *w3t, *b3, *hwt, *hb;  // transposed weights: [in][out]
  const float *s1, *t1, *s2, *t2;               // [nsub][32], [nsub][64]
  float* film;                                  // [B][nsub*192]
  float2* aff1;                                 // [B][nsub][32]
  float2* aff2;                                 // [B][nsub][64]
  int Fd, H, nsub;
};

#ifndef MST_FILM_U
#define MST_FILM_U 4   // iterations of the three dot-product loops in flight (16 loads per thread): the kernel is load-latency-bound
                     // (71 -> 45 us per 72 clips; 8 and 16 spill: 123 us)
#endif
__global__ __launch_bounds__(256) void film_kernel(const FilmParams p) {
  // grid = (clip, group of sub-bands); the two small hidden layers are recomputed per group (82 k MACs)
  extern __shared__ float sm[];
  float* f = sm;            // [Fd]
  float* h1 = f + p.Fd;     // [H]
  float* h2 = h1 + p.H;     // [H]
  float* fl = h2 + p.H;     // [bands_per_group * 192] film outputs of this group
  const int b = blockIdx.x, tid = threadIdx.x;
  const int bpg = (p.nsub + gridDim.y - 1) / gridDim.y;
  const int band0 = blockIdx.y * bpg, band1 = min(p.nsub, band0 + bpg);
  if (band0 >= band1) return;
  for (int i = tid; i < p.Fd; i += 256) f[i] = p.feats[(size_t)b * p.Fd + i];
  __syncthreads();
  for (int j = tid; j < p.H; j += 256) {
    float a0 = p.b0[j], a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
#pragma unroll MST_FILM_U
    for (; i + 3 < p.Fd; i += 4) {
      a0 = fmaf(p.w0t[(size_t)i * p.H + j], f[i], a0);
      a1 = fmaf(p.w0t[(size_t)(i + 1) * p.H + j], f[i + 1], a1);
      a2 = fmaf(p.w0t[(size_t)(i + 2) * p.H + j], f[i + 2], a2);
      a3 = fmaf(p.w0t[(size_t)(i + 3) * p.H + j], f[i + 3], a3);
    }
    for (; i < p.Fd; ++i) a0 = fmaf(p.w0t[(size_t)i * p.H + j], f[i], a0);
    h1[j] = fmaxf((a0 + a1) + (a2 + a3), 0.f);
  }
  __syncthreads();
  for (int j = tid; j < p.H; j += 256) {
    float a0 = p.b3[j], a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
#pragma unroll MST_FILM_U
    for (; i + 3 < p.H; i += 4) {
      a0 = fmaf(p.w3t[(size_t)i * p.H + j], h1[i], a0);
      a1 = fmaf(p.w3t[(size_t)(i + 1) * p.H + j], h1[i + 1], a1);
      a2 = fmaf(p.w3t[(size_t)(i + 2) * p.H + j], h1[i + 2], a2);
      a3 = fmaf(p.w3t[(size_t)(i + 3) * p.H + j], h1[i + 3], a3);
    }
    for (; i < p.H; ++i) a0 = fmaf(p.w3t[(size_t)i * p.H + j], h1[i], a0);
    h2[j] = fmaxf((a0 + a1) + (a2 + a3), 0.f);
  }
  __syncthreads();
  const int nout = p.nsub * 192;
  float* film = p.film + (size_t)b * nout;
  for (int o = band0 * 192 + tid; o < band1 * 192; o += 256) {
    float a0 = p.hb[o], a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
#pragma unroll MST_FILM_U
    for (; i + 3 < p.H; i += 4) {
      a0 = fmaf(p.hwt[(size_t)i * nout + o], h2[i], a0);
      a1 = fmaf(p.hwt[(size_t)(i + 1) * nout + o], h2[i + 1], a1);
      a2 = fmaf(p.hwt[(size_t)(i + 2) * nout + o], h2[i + 2], a2);
      a3 = fmaf(p.hwt[(size_t)(i + 3) * nout + o], h2[i + 3], a3);
    }
    for (; i < p.H; ++i) a0 = fmaf(p.hwt[(size_t)i * nout + o], h2[i], a0);
    const float v = (a0 + a1) + (a2 + a3);
    film[o] = v;
    fl[o - band0 * 192] = v;
  }
  __syncthreads();
  for (int idx = band0 * 96 + tid; idx < band1 * 96; idx += 256) {
    const int band = idx / 96, c = idx % 96;
    const float* fb = fl + (band - band0) * 192;
    if (c < 32) {
      const float g = fb[c], be = fb[32 + c];
      p.aff1[((size_t)b * p.nsub + band) * 32 + c] = make_float2(g * p.s1[band * 32 + c], fmaf(g, p.t1[band * 32 + c], be));
    } else {
      const int c2 = c - 32;
      const float g = fb[64 + c2], be = fb[128 + c2];
      p.aff2[((size_t)b * p.nsub + band) * 64 + c2] =
          make_float2(g * p.s2[band * 64 + c2], fmaf(g, p.t2[band * 64 + c2], be));
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fused conv + affine + ReLU + max-pool
// ------------------------------------------------------------------------------------------
template <int LAYER, int SUB>
struct CC;
template <int SUB>
struct CC<1, SUB> {  // Conv2d(8 -> 32, 7x7, pad 3) + MaxPool2d((SUB, 5)),  SUB = max(1, split // 10) in {1, 2}
  static constexpr int CIN = 8, COUT = 32, NCH = 2, MT = 5, NT = 2;
  static constexpr int WIN = 5 * SUB;       // positions per pooling window
  static constexpr int WPG = 20 / WIN;      // windows per 16-lane group (20 accumulator rows per group)
  static constexpr int TROWS = SUB, TCOLS = 5 * 4 * WPG;  // input positions covered by one wave tile
  static constexpr int PR = TROWS + 6, PC = TCOLS + 6;    // LDS patch (halo 3)
  static constexpr int RL = PC <= 64 ? 64 : 128;          // loader row pitch (power of two >= PC)
};
template <int SUB>
struct CC<2, SUB> {  // Conv2d(32 -> 64, 7x7, pad 3) + MaxPool2d((4, 4)); wave tile = 2x2 windows = 8x8 positions
  static constexpr int CIN = 32, COUT = 64, NCH = 8, MT = 4, NT = 4;
  static constexpr int WIN = 16, WPG = 1;
  static constexpr int TROWS = 8, TCOLS = 8;
  static constexpr int PR = 14, PC = 14;
  static constexpr int RL = 16;
};

template <int SUB>
struct CC<3, SUB> {  // conv2 INPUT gradient: Conv2d(64 -> 32, 7x7, pad 3) of d(conv2 output) with the transposed, flipped
                     // weights; conv1's tile geometry (2 rows x 40 columns of the 10 x 344 plane), raw output
  static constexpr int CIN = 64, COUT = 32, NCH = 16, MT = 5, NT = 2;
  static constexpr int WIN = 10, WPG = 2;
  static constexpr int TROWS = 2, TCOLS = 40;
  static constexpr int PR = TROWS + 6, PC = TCOLS + 6;
  static constexpr int RL = 64;
};

template <int SUB>
struct CC<4, SUB> {  // conv2 (32 -> 64) on conv1's 2 x 40 tile geometry: the training forward's 2-row strip (output rows
                     // 8, 9 of 10 exist only for the batch statistics; an 8 x 8 tile there would be 75 % padding)
  static constexpr int CIN = 32, COUT = 64, NCH = 8, MT = 5, NT = 4;
  static constexpr int WIN = 10, WPG = 2;
  static constexpr int TROWS = 2, TCOLS = 40;
  static constexpr int PR = TROWS + 6, PC = TCOLS + 6;
  static constexpr int RL = 64;
};

struct ConvParams {
  const float* in;
  const float* wfrag;  // [nsub][NCH][WBP]   B fragments: [tap][nt][lane], zero padded to the chunk pitch
  const float2* aff;   // [B][nsub][COUT]
  float* out;
  int B, nsub;
  int in_rows, in_cols;       // valid extent of one band's input plane
  int in_cstride;             // floats between input channels
  int in_bandoff;             // floats between bands
  long long in_clipstride;    // floats between clips
  int out_rows, out_cols;     // pooled plane per (band, channel)
  int tiles_r, tiles_c;       // wave tiles per (band, clip)
  int sets_per_band;          // ceil(B * tiles_r * tiles_c / 8)
  // MODE 1 (training forward): the raw convolution output + bias is stored in ACCUMULATOR ORDER,
  //   yraw[clip][band][tr][tc][n][lane][e]   e < 4*MT (the lane's accumulator slots of N-tile n),
  // which is exactly how the pooling windows sit in registers (the apply / backward kernels work per lane again) and
  // how a later weight-gradient MFMA wants its A operand; per-(band, channel) sums of y and y^2 over the valid
  // positions go to stats[band][COUT][2] (double) for the batch statistics of train-mode BatchNorm.
  float* yraw;
  mst::DetAcc* stats;      // order-independent accumulators (common.h): [nsub][COUT][2]
  const float* bias;          // [nsub][COUT]
  int raw_rows, raw_cols;     // valid extent of the convolution output plane
  // MODE 2 (conv2 input gradient): raw output to out[clip][band][COUT][raw_rows][raw_cols], times the Dropout keep-mask
  const unsigned char* mask;
  float mask_scale;
  // accumulator-order layout of yraw (tile rows / columns of the 8 x 8 or SUB x 40 tiling it is indexed by) and the
  // first output row of this launch (MODE 3: the strip kernel scatters its 2-row tiles into the 8 x 8 layout)
  int acc_tr, acc_tc, row_off;
  // split-precision f16 path: per-(clip, band) power-of-two scale of conv1's pooled output, [B][nsub][2] = (s, 1/s)
  // (f16_scale_kernel); conv1 stores m * s as f16 hi/lo, conv2 folds 1/s into its affine.  NULL = unscaled.
  const float* f16_scale;
  const float* f16_winv;      // [nsub][COUT] inverse of the weight fragments' per-channel pre-scale
  // conv1_resident_kernel<2> only: 1 = MaxPool2d((1, 5)) on its 2 x 40 tiles (16-mel sub-bands: split // 10 == 1) -- the
  // lane's 10 accumulator positions are two 1 x 5 windows, one per tile row, instead of one 2 x 5 window; 0 / 2 = (2, 5)
  int pool_h;
  int clip_major;   // conv1_resident_kernel: order of the sets, see its decode()
  // f16 training (mode 1): the raw output is STORED as float16 times the band's power-of-two scale y_scale[band][2] = (s, 1/s)
  // (yraw16, same accumulator-order layout) -- the conv outputs of the reference's autocast step are half tensors too --
  // and the batch statistics are those of the stored values.  NULL: fp32 yraw.
  _Float16* yraw16;
  const float* y_scale;
  // conv1 with a CHANNEL-MINOR log-mel (mst.h MST_LOGMEL_CM32 / CM16; LAY template argument of the conv1 kernels):
  // `in` (and in_lo) = [B][frames][cm_mels][8 ch]; a band's rows start at mel row band * cm_overlap; in_clipstride is in
  // elements of the layout's 16-byte units' scalar type (floats: frames * cm_mels * 8)
  const void* in_lo;
  int cm_mels, cm_overlap;
};

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
// one lane-unit (NV accumulator slots) of the raw convolution output: fp32, or (y_scale != NULL) float16 times the band's scale
// Order of a lane-unit's NV / 4 four-slot vectors in the raw-output tensors ("accumulator order"): T-MAJOR inside every block of
// 64 lane-units (one (tile, N-tile)): vector t of lane l at [64-block][t][l].  A wave's access to vector t is then one
// contiguous run (512 B of float16, 1 KB of fp32) -- for the convolution's stores, for the element-wise passes that read the
// tensor back and write the gradient over it, and for the weight-gradient kernels that stage it (lane-major, the first
// version, made every such access 64 pieces 40 / 80 bytes apart: five times the cache lines per instruction).
__device__ __forceinline__ size_t yvec(size_t unit, int t, int nvec) {
  return (unit & ~(size_t)63) * nvec + (size_t)t * 64 + (unit & 63);
}
template <int NV>
__device__ __forceinline__ void load_unit(const void* yraw, const float* y_scale, int band, size_t unit, float (&v)[NV]) {
  if (y_scale) {   // kernel-uniform
    const float inv = y_scale[band * 2 + 1];
    const h16x4* s = reinterpret_cast<const h16x4*>(yraw);
#pragma unroll
    for (int t = 0; t < NV / 4; ++t) {
      const h16x4 q = s[yvec(unit, t, NV / 4)];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * t + r] = (float)q[r] * inv;
    }
  } else {
    const f32x4* s = reinterpret_cast<const f32x4*>(yraw);
#pragma unroll
    for (int t = 0; t < NV / 4; ++t) {
      const f32x4 q = s[yvec(unit, t, NV / 4)];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * t + r] = q[r];
    }
  }
}

// fold a lane's running (sum, sum of squares) of N-tile n over the 4 lane groups and add them to stats[band][ch][2]
template <int NT, int COUT>
__device__ __forceinline__ void flush_stats(double (&st)[NT][2], mst::DetAcc* stats, int band, int lane) {
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      double x = st[n][k];
      x += __shfl_xor(x, 16, 64);
      x += __shfl_xor(x, 32, 64);
      if (lane < 16) mst::det_add(&stats[((size_t)band * COUT + n * 16 + lane) * 2 + k], x);
      st[n][k] = 0.0;
    }
}

struct Tile {
  int valid, clip, band, tr, tc;
};

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}


template <int LAYER, int SUB>
struct ConvGeom {
  using C = CC<LAYER, SUB>;
  static constexpr int WCH = 49 * C::NT * 64;                                   // floats per weight chunk
  static constexpr int NWF = (WCH / 4 + kConvThreads - 1) / kConvThreads;      // float4 weight prefetches per thread
  static constexpr int WBP = NWF * kConvThreads * 4;                            // chunk pitch (floats), global and LDS
  static constexpr int PATCH = (4 * C::PR * C::PC + 3) / 4 * 4;                 // floats per wave patch
  static constexpr int NPF = (4 * C::PR * C::RL + 63) / 64;                     // patch prefetches per lane
  static constexpr int NCV = C::RL >= 64 ? C::RL / 64 : 1;                      // column variants of the loader
};

template <int LAYER, int SUB, int MODE = 0>
__global__ __launch_bounds__(kConvThreads) void conv_kernel(const ConvParams p) {
  using C = CC<LAYER, SUB>;
  using GEO = ConvGeom<LAYER, SUB>;
  static_assert(MODE == 0 || (MODE == 1 && LAYER <= 2) || (MODE == 2 && LAYER == 3) || (MODE == 3 && LAYER == 4),
                "raw modes: forward layers, conv2 dgrad, conv2 strip");
  constexpr bool GEO1 = LAYER == 1 || LAYER >= 3;   // 2 x 40 tiles, pooling-window accumulator order
  constexpr int MT = C::MT, NT = C::NT, NCH = C::NCH, PR = C::PR, PC = C::PC, RL = C::RL;
  constexpr int WBP = GEO::WBP, PATCH = GEO::PATCH, NWF = GEO::NWF, NPF = GEO::NPF, NCV = GEO::NCV;
  // BL (conv2): the next chunk is fetched with RAW BUFFER LOADS.  A patch piece is (channel i / 4, row block i % 4): lane = 16 rs + col
  // -> patch row 4 (i % 4) + rs, column col -- 16 pieces instead of 14, but every address is a per-ROW-BLOCK lane offset (computed
  // once per tile; a lane outside the plane gets an offset beyond the buffer and the hardware returns zeros) plus a SCALAR offset
  // per piece: no vector instruction per piece for any tile whose 16 columns lie inside the plane, and no masking when the chunk
  // is staged.  (Per-lane clamped addresses were ~10 VALU instructions per piece; inside an MFMA stream a VALU instruction costs
  // 11-19 cycles of the matrix pipe, scripts/ubench_mfma.hip modes 6, 7.)
  constexpr bool BL = LAYER == 2;
  constexpr int NPFK = BL ? 16 : NPF;
  constexpr int NPIECE = NPFK + NWF, PPT = (NPIECE + 48) / 49;  // prefetch pieces, pieces issued per tap
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: tile bookkeeping stays on the SALU
  float* wbuf = smem;                                  // [2][WBP]
  float* pbuf = smem + 2 * WBP + wave * PATCH;         // wave-private patch [4][PR][PC]

  const int G = gridDim.x;
  const int wg = mst::xcd_remap(blockIdx.x, G);
  const int total_sets = p.nsub * p.sets_per_band;
  // sets are dealt round-robin over the workgroups, except in the modes that accumulate batch statistics (1, 3): there every
  // workgroup takes a CONTIGUOUS range, so that its band -- and the flush of the sums, 128..256 integer atomics per wave --
  // changes once or twice per launch instead of every few tiles (see conv2_f16x3_kernel)
  constexpr bool CONTIG = MODE == 1 || MODE == 3;
  const int s_first = CONTIG ? (int)((long long)wg * total_sets / G) : wg;
  const int my_sets = CONTIG ? (int)((long long)(wg + 1) * total_sets / G) - s_first
                             : (wg < total_sets ? (total_sets - wg + G - 1) / G : 0);
  const int nq = my_sets * NCH;
  const int tpb = p.tiles_r * p.tiles_c;

  auto decode = [&](int q) __attribute__((always_inline)) {
    const int s = CONTIG ? min(s_first + q / NCH, total_sets - 1) : wg + (q / NCH) * G;
    Tile t;
    t.band = min(s / p.sets_per_band, p.nsub - 1);
    const int idx = (s - t.band * p.sets_per_band) * kConvWaves + wave;
    t.valid = (q < nq) && idx < p.B * tpb;
    t.clip = t.valid ? idx / tpb : 0;
    const int ti = t.valid ? idx - t.clip * tpb : 0;
    t.tc = ti / p.tiles_r;
    t.tr = ti - t.tc * p.tiles_r;
    return t;
  };

  // TS (conv2, eval): M-tile t holds tile rows 2 t and 2 t + 1 (lane groups 0, 1 / 2, 3) instead of rows t and t + 4.  The plane's
  // first tile row then has an M-tile (t = 0: rows 0, 1) that sees NOTHING BUT ZERO PADDING through tap rows 0 and 1 (input rows -3
  // .. -1): its 4 x 14 MFMAs per chunk are left out -- 56 of 784, 7 % of the kernel, for every tile when the pooled plane is one
  // tile row high (the contract geometry).  A 4 x 4 pooling window is then spread over the lane pair (l, l ^ 32): one cross-lane
  // max per output.  The raw-output modes keep the accumulator order their consumers index.
  constexpr bool TS = LAYER == 2 && MODE == 0;
  // A-fragment base offsets (floats, inside the wave patch) for this lane: row i = lane & 15 of M-tile t maps to
  // accumulator slot e = 4 t + (i & 3) of lane group g = i >> 2, i.e. window e / WIN, position e % WIN.
  const int kq = lane >> 4, ai = lane & 15, ag = ai >> 2, areg = ai & 3;
  int abase[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int dr, wc;
    if constexpr (GEO1) {
      const int e = 4 * t + areg, wv = e / C::WIN, pos = e % C::WIN;
      dr = pos / 5;
      wc = 5 * (C::WPG * ag + wv) + pos % 5;
    } else {
      dr = TS ? 2 * t + (ag >> 1) : 4 * (ag >> 1) + t;
      wc = 4 * (ag & 1) + areg;
    }
    abase[t] = kq * (PR * PC) + dr * PC + wc;
  }

  // ---- prefetch state of the NEXT chunk: branch-free loads at clamped addresses, masked when staged into LDS
  f32x4 wreg[NWF];
  float pf[NPFK];
  unsigned long long rowmask = 0;   // RL >= 64: bit i = row of piece i is inside the image (wave-uniform)
  unsigned colmask = 0;             // RL >= 64: bit j = column variant j of this lane is inside; RL == 16: bit i = piece i
  int coff[NCV];                    // clamped column offsets of this lane
  const float* nsrc = p.in;         // wave-uniform base of the next tile's 4-channel slab
  const f32x4* nwsrc = reinterpret_cast<const f32x4*>(p.wfrag);
  int nrow0 = 0, ncol0 = 0, nvalid = 0;
  __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, 0, 0x00020000);   // BL: the 4-channel slab
  __amdgpu_buffer_rsrc_t nwrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wfrag), 0, 0, 0x00020000);   // BL: the weight chunk
  unsigned nvoff[4] = {0, 0, 0, 0}, nsoff[4] = {0, 0, 0, 0};   // BL: per row block -- lane offset (bytes, or out of range), scalar offset

  auto prefetch_setup = [&](int q, const Tile& t) __attribute__((always_inline)) {
    const int chunk = q % NCH;
    nwsrc = reinterpret_cast<const f32x4*>(p.wfrag + ((size_t)t.band * NCH + chunk) * WBP);
    nrow0 = (GEO1 ? C::TROWS * t.tr : 8 * t.tr) + (MODE == 3 ? p.row_off : 0) - 3;
    ncol0 = C::TCOLS * t.tc - 3;
    nvalid = t.valid;
    nsrc = p.in + (size_t)t.clip * p.in_clipstride + (size_t)t.band * p.in_bandoff + (size_t)(4 * chunk) * p.in_cstride;
    rowmask = 0, colmask = 0;
    if constexpr (BL) {
      nrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(nsrc), 0, (unsigned)(4 * p.in_cstride) * 4u, 0x00020000);
      nwrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(nwsrc), 0, (unsigned)WBP * 4u, 0x00020000);
      if (chunk == 0) {   // the lane offsets belong to the TILE: once per NCH chunks (between chunks only the two scalar bases move)
        const int rs = lane >> 4, col = lane & 15, cin = ncol0 + col;
        const bool col_in = nvalid && col < PC && cin >= 0 && cin < p.in_cols;
        const int cbase = max(ncol0, 0);                               // scalar part of the column (>= 0: soffset is unsigned)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const int rbase = max(nrow0 + 4 * rb, 0), rin = nrow0 + 4 * rb + rs;
          const bool ok = col_in && 4 * rb + rs < PR && rin >= 0 && rin < p.in_rows;
          nsoff[rb] = (unsigned)(rbase * p.in_cols + cbase) * 4u;
          nvoff[rb] = ok ? (unsigned)((rin - rbase) * p.in_cols + (cin - cbase)) * 4u : 0x80000000u;
        }
      }
    }
    if constexpr (RL >= 64) {
#pragma unroll
      for (int j = 0; j < NCV; ++j) {
        const int col = lane + 64 * j, cin = ncol0 + col;
        if (col < PC && cin >= 0 && cin < p.in_cols) colmask |= 1u << j;
        coff[j] = min(max(cin, 0), p.in_cols - 1);
      }
    }
  };
  auto prefetch_piece = [&](int i) __attribute__((always_inline)) {  // i is a compile-time constant after unrolling
    if constexpr (BL) {
      if (i < NPFK) {
        pf[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(nrsrc, nvoff[i & 3], nsoff[i & 3] + (unsigned)((i >> 2) * p.in_cstride) * 4u, 0));
      } else if (i < NPIECE) {
        wreg[i - NPFK] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(nwrsrc, (unsigned)tid * 16u, (unsigned)(kConvThreads * (i - NPFK)) * 16u, 0));
      }
    } else
    if (i < NPF) {
      if constexpr (RL >= 64) {
        const int row = i / NCV, j = i % NCV;
        const int cc = row / PR, r = row % PR;
        const int rin = nrow0 + r;
        const int rc = min(max(rin, 0), p.in_rows - 1);
        if (nvalid && rin == rc) rowmask |= 1ull << i;
        const float* rowp = nsrc + (size_t)cc * p.in_cstride + (size_t)rc * p.in_cols;  // wave-uniform
        pf[i] = rowp[coff[j]];
      } else {
        const int e = lane + 64 * i;
        const int row = e / RL, col = e % RL;
        const int cc = min(row / PR, 3), r = row % PR;
        const int rin = nrow0 + r, cin = ncol0 + col;
        const int rc = min(max(rin, 0), p.in_rows - 1), cl = min(max(cin, 0), p.in_cols - 1);
        if (nvalid && row < 4 * PR && col < PC && rin == rc && cin == cl) colmask |= 1u << i;
        pf[i] = nsrc[cc * p.in_cstride + rc * p.in_cols + cl];
      }
    } else if (i < NPIECE) {
      wreg[i - NPF] = nwsrc[tid + kConvThreads * (i - NPF)];
    }
  };

  f32x4 acc[MT][NT];
  double st[NT][2];
  int st_band = -1;
#pragma unroll
  for (int n = 0; n < NT; ++n) st[n][0] = st[n][1] = 0.0;
  Tile cur{}, nxt{};
  nxt = decode(0);
  if (nq > 0) {
    prefetch_setup(0, nxt);
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) prefetch_piece(i);
  }
#ifdef MST_TRACE
  long long trc[5] = {0, 0, 0, 0, 0};   // cycles: staging, barrier wait, taps, -; chunks
  const long long trc_start = clock64();
#endif
  for (int q = 0; q < nq; ++q) {
    float* wb = wbuf + (q & 1) * WBP;
#ifdef MST_TRACE
    const long long tq0 = clock64();
#endif
    // stage the prefetched chunk into LDS (unconditional stores; out-of-image elements become zeros)
#pragma unroll
    for (int i = 0; i < NWF; ++i) reinterpret_cast<f32x4*>(wb)[tid + kConvThreads * i] = wreg[i];
    if constexpr (BL) {
#pragma unroll
      for (int i = 0; i < NPFK; ++i) {
        const int rs = lane >> 4, col = lane & 15, r = 4 * (i & 3) + rs;
        if (r < PR && col < PC) pbuf[((i >> 2) * PR + r) * PC + col] = pf[i];   // (out-of-plane elements were loaded as zeros)
      }
    } else
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      if constexpr (RL >= 64) {
        const int row = i / NCV, j = i % NCV, col = lane + 64 * j;
        const bool ok = ((rowmask >> i) & 1ull) && ((colmask >> j) & 1u);
        if (col < PC) pbuf[row * PC + col] = ok ? pf[i] : 0.f;
      } else {
        const int e = lane + 64 * i;
        const int row = e / RL, col = e % RL;
        if (row < 4 * PR && col < PC) pbuf[row * PC + col] = ((colmask >> i) & 1u) ? pf[i] : 0.f;
      }
    }
#ifdef MST_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long tq1 = clock64();
#endif
    __syncthreads();
#ifdef MST_TRACE
    const long long tq2 = clock64();
    trc[0] += tq1 - tq0, trc[1] += tq2 - tq1, trc[4] += 1;
#endif
    cur = nxt;
    if (!BL || (q + 1) % NCH == 0) nxt = decode(q + 1);   // (BL: the tile -- three integer divisions -- changes every NCH chunks only)
    prefetch_setup(q + 1, nxt);   // past the end this re-reads valid memory and is never staged
    const int chunk = q % NCH;
    if (chunk == 0) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // 49 taps x (MT A-fragments, NT B-fragments, MT*NT MFMAs); the fragments of tap+1 are read from LDS and PPT
    // pieces of the next chunk are fetched from global memory while the MFMAs of this tap are in flight
    {
      float a[2][MT], b[2][NT];
#pragma unroll
      for (int t = 0; t < MT; ++t) a[0][t] = pbuf[abase[t]];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[0][n] = wb[n * 64 + lane];
      auto taps = [&](auto LO_, auto HI_, auto SKIP_) __attribute__((always_inline)) {
        constexpr int LO = decltype(LO_)::value, HI = decltype(HI_)::value;
        constexpr bool SKIP0 = decltype(SKIP_)::value;   // M-tile 0 multiplies zeros in these taps: no MFMAs for it
#pragma unroll
        for (int tap = LO; tap < HI; ++tap) {
          const int cu = tap & 1, nx = cu ^ 1;
          const int off = ((tap + 1) / 7) * PC + ((tap + 1) % 7);
          // hand-interleaved issue order, pinned by sched_barrier: after every MFMA one LDS read of the next tap's
          // fragments (or a global prefetch load of the next chunk) is issued into the shadow of that MFMA
#pragma unroll
          for (int i = 0; i < MT * NT; ++i) {
            const int t = i / NT, n = i % NT;
            // (no per-MFMA skipping of M-tiles past the plane: a wave-uniform branch here cost the training forward half its
            // speed; the last rows of a 10-row plane go through the strip kernel, MODE 3, and otherwise padding rows are computed)
            if (!(SKIP0 && t == 0)) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cu][t], b[cu][n], acc[t][n], 0, 0, 0);
            if (tap + 1 < 49) {
              if (i < NT) b[nx][i] = wb[((tap + 1) * NT + i) * 64 + lane];   // B first: needed by the next tap's MFMA 0
              else if (i < MT + NT) a[nx][i - NT] = pbuf[abase[i - NT] + off];
            }
            if (i >= MT + NT && i - (MT + NT) < PPT) prefetch_piece(tap * PPT + (i - (MT + NT)));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      using I0 = std::integral_constant<int, 0>;
      using I14 = std::integral_constant<int, 14>;
      using I49 = std::integral_constant<int, 49>;
      if constexpr (TS) {   // tap rows 0 and 1 (taps 0 .. 13), with or without M-tile 0; then the rest
        if (cur.tr == 0) taps(I0{}, I14{}, std::true_type{});
        else taps(I0{}, I14{}, std::false_type{});
        taps(I14{}, I49{}, std::false_type{});
      } else {
        taps(I0{}, I49{}, std::false_type{});
      }
    }
#ifdef MST_TRACE
    trc[2] += clock64() - tq2;
#endif
    if (MODE == 1 && chunk == NCH - 1 && cur.valid) {
      // training forward: raw output + bias in accumulator order, batch-statistics sums over the valid positions
      const int j = lane & 15, g = lane >> 4;
      if (cur.band != st_band) {
        if (st_band >= 0) flush_stats<NT, C::COUT>(st, p.stats, st_band, lane);
        st_band = cur.band;
      }
      float* yb = p.yraw + ((((size_t)cur.clip * p.nsub + cur.band) * p.acc_tr + cur.tr) * p.acc_tc + cur.tc) *
                               (size_t)(NT * 64 * 4 * MT);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float b = p.bias[cur.band * C::COUT + n * 16 + j];
        f32x4* dst = reinterpret_cast<f32x4*>(yb) + (size_t)n * (64 * MT) + lane;   // t-major (yvec): vector t at dst[64 t]
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          f32x4 v = acc[t][n];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] += b;
            int row, col;
            if constexpr (LAYER == 1) {
              const int e = 4 * t + r, wv = e / C::WIN, pos = e % C::WIN;
              row = C::TROWS * cur.tr + pos / 5, col = C::TCOLS * cur.tc + 5 * (C::WPG * g + wv) + pos % 5;
            } else {
              row = 8 * cur.tr + 4 * (g >> 1) + t, col = 8 * cur.tc + 4 * (g & 1) + r;
            }
            if (row < p.raw_rows && col < p.raw_cols) st[n][0] += (double)v[r], st[n][1] += (double)v[r] * (double)v[r];
          }
          dst[64 * t] = v;
        }
      }
    }
    if (MODE == 3 && chunk == NCH - 1 && cur.valid) {
      // conv2 strip (training forward): raw output + bias scattered into the 8 x 8-tile accumulator layout the
      // downstream kernels index, batch-statistics sums over the valid positions
      const int j = lane & 15, g = lane >> 4;
      if (cur.band != st_band) {
        if (st_band >= 0) flush_stats<NT, C::COUT>(st, p.stats, st_band, lane);
        st_band = cur.band;
      }
      float* yplane = p.yraw + ((size_t)cur.clip * p.nsub + cur.band) * p.acc_tr * p.acc_tc * (size_t)(NT * 64 * 16);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float b = p.bias[cur.band * C::COUT + n * 16 + j];
#pragma unroll
        for (int e = 0; e < 4 * MT; ++e) {
          const int wv = e / C::WIN, pos = e % C::WIN;
          const int row = p.row_off + C::TROWS * cur.tr + pos / 5, col = C::TCOLS * cur.tc + 5 * (C::WPG * g + wv) + pos % 5;
          if (row < p.raw_rows && col < p.raw_cols) {
            const float v = acc[e >> 2][n][e & 3] + b;
            st[n][0] += (double)v, st[n][1] += (double)v * (double)v;
            const int tr8 = row >> 3, rr = row & 7, tc8 = col >> 3, cc8 = col & 7;
            const int lane8 = 16 * (((rr >> 2) << 1) | (cc8 >> 2)) + j, e8 = 4 * (rr & 3) + (cc8 & 3);
            yplane[(((((size_t)tr8 * p.acc_tc + tc8) * NT + n) * 4 + (e8 >> 2)) * 64 + lane8) * 4 + (e8 & 3)] = v;   // t-major (yvec)
          }
        }
      }
    }
    if (MODE == 2 && chunk == NCH - 1 && cur.valid) {
      // conv2 input gradient: raw values to the pool1-shaped tensor, Dropout mask applied on the way out
      const int j = lane & 15, g = lane >> 4;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int ch = n * 16 + j;
        float* ob = p.out + (((size_t)cur.clip * p.nsub + cur.band) * C::COUT + ch) * (size_t)p.raw_rows * p.raw_cols;
        const unsigned char* mb = p.mask ? p.mask + (((size_t)cur.clip * p.nsub + cur.band) * C::COUT + ch) * (size_t)p.raw_rows * p.raw_cols
                                         : nullptr;
#pragma unroll
        for (int e = 0; e < 4 * MT; ++e) {
          const int wv = e / C::WIN, pos = e % C::WIN;
          const int row = C::TROWS * cur.tr + pos / 5, col = C::TCOLS * cur.tc + 5 * (C::WPG * g + wv) + pos % 5;
          if (row < p.raw_rows && col < p.raw_cols) {
            const size_t o = (size_t)row * p.raw_cols + col;
            float v = acc[e >> 2][n][e & 3];
            if (mb) v = mb[o] ? v * p.mask_scale : 0.f;
            ob[o] = v;
          }
        }
      }
    }
    if (MODE == 0 && chunk == NCH - 1 && cur.valid) {
      // epilogue: y = A*acc + C, ReLU, max over the window, all in this lane's registers
      const int j = lane & 15, g = lane >> 4;
      const float2* aff = p.aff + ((size_t)cur.clip * p.nsub + cur.band) * C::COUT;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int ch = n * 16 + j;
        const float2 ac = aff[ch];
        if constexpr (LAYER == 1) {
#pragma unroll
          for (int wv = 0; wv < C::WPG; ++wv) {
            float m = 0.f;  // ReLU floor
#pragma unroll
            for (int pos = 0; pos < C::WIN; ++pos) {
              const int e = wv * C::WIN + pos;
              m = fmaxf(m, fmaf(acc[e >> 2][n][e & 3], ac.x, ac.y));
            }
            const int pc = 4 * C::WPG * cur.tc + C::WPG * g + wv;
            if (pc < p.out_cols)
              p.out[((((size_t)cur.clip * p.nsub + cur.band) * C::COUT + ch) * p.out_rows + cur.tr) * p.out_cols + pc] = m;
          }
        } else {
          float m = 0.f;
          if constexpr (TS) {   // rows 2 t + (g >> 1): pooled row 0 = M-tiles 0, 1 of the lane pair (l, l ^ 32), pooled row 1 = M-tiles 2, 3
            float m0 = 0.f, m1 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              m0 = fmaxf(m0, fmaxf(fmaf(acc[0][n][r], ac.x, ac.y), fmaf(acc[1][n][r], ac.x, ac.y)));
              m1 = fmaxf(m1, fmaxf(fmaf(acc[2][n][r], ac.x, ac.y), fmaf(acc[3][n][r], ac.x, ac.y)));
            }
            m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, 32, 64));
            m = (g >> 1) ? m1 : m0;
          } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) m = fmaxf(m, fmaf(acc[t][n][r], ac.x, ac.y));
          }
          const int pr = 2 * cur.tr + (g >> 1), pc = 2 * cur.tc + (g & 1);
          if (pr < p.out_rows && pc < p.out_cols)  // pool_in[clip][(band*64 + ch)*FD + pr][pc]
            p.out[(((size_t)cur.clip * p.nsub + cur.band) * C::COUT + ch) * p.out_rows * p.out_cols +
                  (size_t)pr * p.out_cols + pc] = m;
        }
      }
    }
  }
#ifdef MST_TRACE
  if (MODE == 0 && LAYER == 2 && lane == 0 && p.yraw) {
    float* o = p.yraw + (blockIdx.x * kConvWaves + wave) * 8;
    for (int i = 0; i < 5; ++i) o[i] = (float)trc[i];
    o[5] = (float)(clock64() - trc_start);
  }
#endif
  if ((MODE == 1 || MODE == 3) && st_band >= 0) flush_stats<NT, C::COUT>(st, p.stats, st_band, lane);
}

// ------------------------------------------------------------------------------------------
// conv1 with band-resident weights (SUB == 2, the reference default 20-mel sub-bands).
// All 392 x 32 weights of one sub-band (50 KB) stay in LDS next to eight wave-private 8-channel patches
// (8 x 11.8 KB): there is NO per-tile barrier.  Every wave free-runs over its own tiles, so the staging / epilogue
// of one wave overlaps the MFMAs of its SIMD partner.  Workgroups own contiguous runs of sets, so the band (and
// with it the LDS weight image) changes at most once per workgroup.
// ------------------------------------------------------------------------------------------
// LAY 0: log-mel in the reference layout (B, 8, M, F): one 64-lane row load per (channel, patch row), 64 per tile.
// LAY 1: channel-minor log-mel [B][F][M][8] (MST_LOGMEL_CM32): a frame of the patch -- 8 mel rows x 8 channels -- is 256
//        contiguous bytes; a lane loads 16-byte units (frame, row, half = 4 channels), 12 per tile, every lane busy, and
//        scatters them into the same channel-major LDS patch (4 ds_write_b32 per unit, 2-way bank conflicts at most).
template <int SUB, int MODE = 0, int LAY = 0>
__global__ __launch_bounds__(kConvThreads) void conv1_resident_kernel(const ConvParams p) {
  using C = CC<1, SUB>;
  constexpr int MT = C::MT, NT = C::NT, PR = C::PR, PC = C::PC;
  static_assert(C::RL == 64, "resident conv1 expects one loader row per instruction");
  static_assert(LAY == 0 || PR == 8, "the channel-minor loader maps lane bits 1..3 to the 8 patch rows");
  constexpr int WCH = 49 * NT * 64;                  // floats per 4-channel weight chunk
  constexpr int WBP = ConvGeom<1, SUB>::WBP;         // chunk pitch in global memory
  constexpr int CHS = PR * PC;                       // floats per patch channel
  constexpr int PATCH = 8 * CHS;                     // 8 input channels
  constexpr int NPF = LAY == 0 ? 8 * PR : (PC + 3) / 4;   // LAY 0: one 64-lane row load per (channel, row); LAY 1: 4 frames per load
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* wres = smem;                                // [2][WCH]
  float* pbuf = smem + 2 * WCH + wave * PATCH;
  constexpr bool DYN = MODE == 0;                    // tiles by ticket (below); the training forward keeps the static deal: its
                                                     // per-wave statistics sums are folded in a fixed order
  constexpr int kTickets = 16;
  __shared__ unsigned tickets[kTickets];
#ifdef MST_TRACE
  long long trc[5] = {0, 0, 0, 0, 0};   // cycles: k-steps, epilogue, stage + draw + setup, segment barrier + weights; tiles
  const long long trc_start = clock64();
#endif

  const int G = gridDim.x;
  const int wg = mst::xcd_remap(blockIdx.x, G);
  // Order of the sets.  Band-major (clip_major = 0; the training forward: the batch statistics are flushed at every band change,
  // so a workgroup should see as few as possible): sets_per_band sets per band, the clips' tiles concatenated inside it.
  // Clip-group-major (clip_major = k > 0; the eval forward): the clips are taken k at a time, and inside a group the bands follow
  // each other ([group][band][clip of the group][tile]; sets_per_band counts the sets of ONE (group, band)).  Measured on 72
  // clips (FETCH_SIZE x 2 per launch, kernel time equal within +-0.5 % for all of them): band-major 1.93 GB, k = 4 1.77 GB,
  // k = 1 1.11 GB = 72 x 11 patches of 26 mel rows, i.e. every (clip, band) patch fetched exactly once; the sub-bands' shared
  // rows (each mel row sits in two bands: 0.51 GB algorithmic) are never in L2 long enough in any order -- a workgroup meets them
  // again a whole (clip, band) later.  k = 1 pads the 430 tiles of a (clip, band) to 54 sets (0.5 % empty wave slots).
  const int total_sets = (p.clip_major ? (p.B + p.clip_major - 1) / p.clip_major : 1) * p.nsub * p.sets_per_band;
  const int s_begin = (int)((long long)wg * total_sets / G), s_end = (int)((long long)(wg + 1) * total_sets / G);
  const int tpb = p.tiles_r * p.tiles_c;

  auto decode = [&](int s) __attribute__((always_inline)) {
    Tile t;
    int ti;
    if (p.clip_major) {   // groups of clip_major clips: [group][band][clip of the group][tile]
      const int ngrp = (p.B + p.clip_major - 1) / p.clip_major;
      const int gb = min(s / p.sets_per_band, ngrp * p.nsub - 1);
      const int idx = (s - gb * p.sets_per_band) * kConvWaves + wave;
      const int grp = gb / p.nsub, c0 = grp * p.clip_major;
      t.band = gb - grp * p.nsub;
      t.valid = (s < s_end) && idx < min(p.clip_major, p.B - c0) * tpb;
      const int cl = t.valid ? idx / tpb : 0;
      t.clip = c0 + cl;
      ti = t.valid ? idx - cl * tpb : 0;
    } else {
      t.band = min(s / p.sets_per_band, p.nsub - 1);
      const int idx = (s - t.band * p.sets_per_band) * kConvWaves + wave;
      t.valid = (s < s_end) && idx < p.B * tpb;
      t.clip = t.valid ? idx / tpb : 0;
      ti = t.valid ? idx - t.clip * tpb : 0;
    }
    t.tc = ti / p.tiles_r;
    t.tr = ti - t.tc * p.tiles_r;
    return t;
  };

  const int kq = lane >> 4, ai = lane & 15, ag = ai >> 2, areg = ai & 3;
  int abase[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int e = 4 * t + areg, wv = e / C::WIN, pos = e % C::WIN;
    abase[t] = kq * CHS + (pos / 5) * PC + 5 * (C::WPG * ag + wv) + pos % 5;
  }

  typename std::conditional<LAY == 0, float, f32x4>::type pf[NPF];
  unsigned long long rowmask = 0;   // LAY 0: bit i = row of piece i inside the band (wave-uniform); LAY 1: bit i = this lane's unit of piece i inside
  bool col_ok = false;              // LAY 0: this lane's column inside the clip; LAY 1: this lane's row inside the band
  int coff = 0, nrow0 = 0, nvalid = 0;
  const float* nsrc = p.in;
  bool nfast = false;                       // DYN, LAY 1: the next tile takes the buffer-load path
  __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, 0, 0x00020000);
  unsigned nsoff = 0, nvoff = 0;
  // DYN: a unit outside the image is LOADED AS ZEROS -- from the zero padding behind a weight chunk (WCH .. WBP of wfrag) -- instead
  // of being masked after the load: staging the patch is then LDS stores alone
  static_assert(WBP - WCH >= 4, "the chunk pitch leaves at least 16 zero bytes behind the weights");
  const float* const zero16 = p.wfrag + WCH;
  // LAY 1: lane = 16 fq + 2 r + half -> patch row r, channels 4 half .. 4 half + 3, frames fq + 4 i (piece i)
  const int lr = (lane >> 1) & 7, lhalf = lane & 1, lfq = lane >> 4;
  const unsigned frame_bytes = (unsigned)p.cm_mels * 32u;
  auto prefetch_setup = [&](const Tile& t) __attribute__((always_inline)) {
    nrow0 = C::TROWS * t.tr - 3;
    nvalid = t.valid;
    rowmask = 0;
    if constexpr (LAY == 0) {
      const int col0 = C::TCOLS * t.tc - 3, cin = col0 + lane;
      nsrc = p.in + (size_t)t.clip * p.in_clipstride + (size_t)t.band * p.in_bandoff;
      col_ok = lane < PC && cin >= 0 && cin < p.in_cols;
      coff = min(max(cin, 0), p.in_cols - 1);
    } else {
      const int rin = nrow0 + lr, rc = min(max(rin, 0), p.in_rows - 1);
      nsrc = p.in + (size_t)t.clip * p.in_clipstride;                       // wave-uniform clip base
      col_ok = nvalid && rin == rc;
      coff = ((t.band * p.cm_overlap + rc) * 8 + 4 * lhalf) * 4;             // byte offset of this lane's (row, half) inside a frame
      nrow0 = C::TCOLS * t.tc - 3 + lfq;                                     // (re-used: first frame of this lane)
      if constexpr (DYN) {
        // Fast path (every tile whose 48 frames lie inside the clip: all but the first and last tile column): the pieces are RAW
        // BUFFER LOADS -- per piece one scalar add (the frame offset, in soffset) and the load; the lane's offset inside a frame
        // is computed ONCE per tile and a lane whose mel row lies outside the band gets an offset beyond the buffer, which the
        // hardware answers with zeros (scripts/ubench_bufload.hip).  Per-lane 64-bit address arithmetic for 12 pieces was ~150
        // VALU instructions per tile, and a VALU instruction inside an MFMA stream costs 11-19 cycles of the matrix pipe
        // (scripts/ubench_mfma.hip modes 6, 7).
        const int f0 = C::TCOLS * t.tc - 3;
        nfast = nvalid && f0 >= 0 && f0 + 4 * NPF + 3 < p.in_cols;
        nrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(nsrc), 0, (unsigned)p.in_cols * frame_bytes, 0x00020000);
        nsoff = (unsigned)max(f0, 0) * frame_bytes;
        nvoff = col_ok ? (unsigned)coff + (unsigned)lfq * frame_bytes : 0x80000000u;
      }
    }
  };
  auto prefetch_piece = [&](int i) __attribute__((always_inline)) {
    if (i < NPF) {
      if constexpr (LAY == 0) {
        const int cc = i / PR, r = i % PR;
        const int rin = nrow0 + r;
        const int rc = min(max(rin, 0), p.in_rows - 1);
        if (nvalid && rin == rc) rowmask |= 1ull << i;
        const float* rowp = nsrc + (size_t)cc * p.in_cstride + (size_t)rc * p.in_cols;
        if constexpr (DYN) pf[i] = (nvalid && rin == rc && col_ok) ? rowp[coff] : *zero16;   // (a select of the ADDRESS)
        else pf[i] = rowp[coff];
      } else {
        const int fin = nrow0 + 4 * i, fc = min(max(fin, 0), p.in_cols - 1);
        if (col_ok && fin == fc) rowmask |= 1ull << i;
        if (DYN && nfast) {
          pf[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(nrsrc, nvoff, nsoff + (unsigned)(4 * i) * frame_bytes, 0));
        } else {
          const unsigned boff = (unsigned)fc * frame_bytes + (unsigned)coff;   // 32-bit offset from the scalar clip base
          const char* src = reinterpret_cast<const char*>(nsrc) + boff;
          if constexpr (DYN) src = (col_ok && fin == fc) ? src : reinterpret_cast<const char*>(zero16);
          pf[i] = *reinterpret_cast<const f32x4*>(src);
        }
      }
    }
  };
  // pieces of the next tile are fetched between the MFMAs of the k-steps: LAY 0 one row per step for the first 64 steps,
  // LAY 1 one 16-byte unit every 8th step
  auto prefetch_step = [&](int ks) __attribute__((always_inline)) {
    if constexpr (LAY == 0) prefetch_piece(ks);
    else if (ks % 8 == 0) prefetch_piece(ks / 8);
  };

  f32x4 acc[MT][NT];
  double st[NT][2];
#pragma unroll
  for (int n = 0; n < NT; ++n) st[n][0] = st[n][1] = 0.0;
  int cur_band = -1;
  // DYN: the workgroup's run of sets [s_begin, s_end) cut at the (group, band) boundaries into segments; inside a segment the
  // tiles go to whichever wave asks next (one LDS counter per segment, kTickets of them in rotation: a wave can be at most
  // seven exhausted segments ahead of the slowest wave, each of the other seven holding at most one tile it has not computed)
  const int spb = p.sets_per_band;
  const int gb0 = s_begin / spb, nseg = s_end > s_begin ? (s_end - 1) / spb - gb0 + 1 : 0;
  int dseg = 0, dlo = 0, dhi = 0, dgb = 0, nxt_seg = 0;
  auto seg_range = [&](int k) __attribute__((always_inline)) {
    dgb = gb0 + k;
    const int cm = p.clip_major ? p.clip_major : p.B, grp = p.clip_major ? dgb / p.nsub : 0;
    dlo = max(s_begin - dgb * spb, 0) * kConvWaves;
    dhi = min((min(s_end, (dgb + 1) * spb) - dgb * spb) * kConvWaves, min(cm, p.B - grp * cm) * tpb);
  };
  // A ticket is ASKED FOR (one LDS atomic by lane 0) several k-steps before it is looked at, so its round trip is never waited for
  unsigned tkv = 0;
  auto ticket_issue = [&]() __attribute__((always_inline)) {
    tkv = 0;
    if (dseg < nseg && lane == 0)
      tkv = __hip_atomic_fetch_add(&tickets[dseg % kTickets], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto ticket_take = [&]() __attribute__((always_inline)) {
    Tile t{};
    while (dseg < nseg) {
      const int idx = dlo + (int)__builtin_amdgcn_readfirstlane(tkv);
      if (idx < dhi) {
        const int grp = p.clip_major ? dgb / p.nsub : 0;
        const int cl = idx / tpb, ti = idx - cl * tpb;
        t.valid = 1;
        t.band = dgb - grp * p.nsub;
        t.clip = grp * p.clip_major + cl;
        t.tc = ti / p.tiles_r;
        t.tr = ti - t.tc * p.tiles_r;
        return t;
      }
      if (++dseg < nseg) seg_range(dseg);   // this segment is empty: on to the next one (a handful of times per launch)
      ticket_issue();
    }
    return t;   // the run is empty
  };
  Tile cur{}, nxt{};
  if constexpr (DYN) {
    if (tid < kTickets) tickets[tid] = 0;
    __syncthreads();
    if (nseg > 0) seg_range(0);
    ticket_issue();
    nxt = ticket_take();
    nxt_seg = dseg;
  } else {
    nxt = decode(s_begin);
  }
  prefetch_setup(nxt);
#pragma unroll
  for (int i = 0; i < NPF; ++i) prefetch_piece(i);

  // stage this wave's prefetched patch (wave-private: no barrier)
  // out-of-image elements of a fetched piece become zeros (the loads went to clamped addresses)
  auto mask_piece = [&](int i) __attribute__((always_inline)) {
    if constexpr (LAY == 0) {
      pf[i] = (((rowmask >> i) & 1ull) && col_ok) ? pf[i] : 0.f;
    } else {
      const bool ok = (rowmask >> i) & 1ull;
#pragma unroll
      for (int k = 0; k < 4; ++k) pf[i][k] = ok ? pf[i][k] : 0.f;
    }
  };
  // MASKED: the pieces are masked already (DYN: inside the k-steps) and this is LDS stores only
  auto stage_ = [&](auto MASKED_) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(MASKED_)::value;
    if constexpr (LAY == 0) {
#pragma unroll
      for (int i = 0; i < NPF; ++i) {
        if (!MASKED) mask_piece(i);
        if (lane < PC) pbuf[i * PC + lane] = pf[i];
      }
    } else {
      float* pl = pbuf + (4 * lhalf) * CHS + lr * PC + lfq;
#pragma unroll
      for (int i = 0; i < NPF; ++i) {
        if (4 * i + 3 < PC || 4 * i + lfq < PC) {
          if (!MASKED) mask_piece(i);
#pragma unroll
          for (int k = 0; k < 4; ++k) pl[k * CHS + 4 * i] = pf[i][k];
        }
      }
    }
  };
  auto stage = [&]() __attribute__((always_inline)) { stage_(std::false_type{}); };
  // one tile: k-steps (the next tile's patch is fetched meanwhile), then the epilogue
  auto tile_body = [&]() __attribute__((always_inline)) {
#ifdef MST_TRACE
    const long long tr0 = clock64();
#endif
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // 98 k-steps = 49 taps x 2 channel groups; fragments of step+1 are read while the MFMAs of step run, and one
      // row of the NEXT tile's patch is fetched from global memory per step for the first 64 steps
      float a[2][MT], b[2][NT];
#pragma unroll
      for (int t = 0; t < MT; ++t) a[0][t] = pbuf[abase[t]];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[0][n] = wres[n * 64 + lane];
#pragma unroll
      for (int ks = 0; ks < 98; ++ks) {
        const int cu = ks & 1, nx = cu ^ 1;
        const int tap = (ks + 1) >> 1, ch = (ks + 1) & 1;
        const int off = ch * 4 * CHS + (tap / 7) * PC + (tap % 7);
#pragma unroll
        for (int i = 0; i < MT * NT; ++i) {
          const int t = i / NT, n = i % NT;
          acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cu][t], b[cu][n], acc[t][n], 0, 0, 0);
          if (ks + 1 < 98) {
            if (i < NT) b[nx][i] = wres[ch * WCH + (tap * NT + i) * 64 + lane];
            else if (i < MT + NT) a[nx][i - NT] = pbuf[abase[i - NT] + off];
          }
          if (i == MT + NT) prefetch_step(ks);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#ifdef MST_TRACE
    const long long tr1 = clock64();
    trc[0] += tr1 - tr0, trc[4] += 1;
#endif
    if constexpr (MODE == 1) {
      // training forward: raw output + bias in accumulator order, batch-statistics sums over the valid columns
      const int j = lane & 15, g = lane >> 4;
      float* yb = p.yraw + ((((size_t)cur.clip * p.nsub + cur.band) * p.acc_tr + cur.tr) * p.acc_tc + cur.tc) *
                               (size_t)(NT * 64 * 4 * MT);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float b = p.bias[cur.band * C::COUT + n * 16 + j];
        f32x4* dst = reinterpret_cast<f32x4*>(yb) + (size_t)n * (64 * MT) + lane;   // t-major (yvec): vector t at dst[64 t]
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          f32x4 v = acc[t][n];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] += b;
            const int e = 4 * t + r, wv = e / C::WIN, pos = e % C::WIN;
            const int col = C::TCOLS * cur.tc + 5 * (C::WPG * g + wv) + pos % 5;
            if (col < p.raw_cols) st[n][0] += (double)v[r], st[n][1] += (double)v[r] * (double)v[r];
          }
          dst[64 * t] = v;
        }
      }
    } else
    {
      const int j = lane & 15, g = lane >> 4;
      const float2* aff = p.aff + ((size_t)cur.clip * p.nsub + cur.band) * C::COUT;
      float* orow = p.out + ((size_t)cur.clip * p.nsub + cur.band) * C::COUT * p.out_rows * p.out_cols +
                    (size_t)cur.tr * p.out_cols;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int ch = n * 16 + j;
        const float2 ac = aff[ch];
#pragma unroll
        for (int wv = 0; wv < C::WPG; ++wv) {
          const int pc = 4 * C::WPG * cur.tc + C::WPG * g + wv;
          if (SUB == 2 && p.pool_h == 1) {   // two 1 x 5 windows: positions 0..4 are tile row 0, 5..9 tile row 1
#pragma unroll
            for (int hr = 0; hr < 2; ++hr) {
              float m = 0.f;
#pragma unroll
              for (int pos = 0; pos < 5; ++pos) {
                const int e = wv * C::WIN + 5 * hr + pos;
                m = fmaxf(m, fmaf(acc[e >> 2][n][e & 3], ac.x, ac.y));
              }
              if (pc < p.out_cols && 2 * cur.tr + hr < p.out_rows)
                orow[(size_t)ch * p.out_rows * p.out_cols + (size_t)(cur.tr + hr) * p.out_cols + pc] = m;   // orow is at row tr: + tr + hr
            }
          } else {
            float m = 0.f;
#pragma unroll
            for (int pos = 0; pos < C::WIN; ++pos) {
              const int e = wv * C::WIN + pos;
              m = fmaxf(m, fmaf(acc[e >> 2][n][e & 3], ac.x, ac.y));
            }
            if (pc < p.out_cols) orow[(size_t)ch * p.out_rows * p.out_cols + pc] = m;
          }
        }
      }
    }
#ifdef MST_TRACE
    trc[1] += clock64() - tr1;
#endif
  };
  // ---- DYN (eval forward): everything of a tile that is not a matrix instruction or the staging of its patch rides INSIDE the
  // k-steps, in the issue slots between the MFMAs: the ticket for the next tile, its decode and prefetch set-up, the zero-fill of
  // the fetched pieces, and the EPILOGUE OF THE PREVIOUS TILE (its accumulators are copied aside when its k-steps end).
  // Why: a wave outside its k-steps shares the SIMD with one that is inside them, and VALU instructions of such a wave wait for
  // gaps in the partner's MFMA stream (scripts/ubench_corun.hip: next to a saturating MFMA wave a v_cndmask / v_fma takes
  // 1700-2300 cycles, a ds_write_b32 20).  Traced at the contract size (profiles/r04_conv1_wave_trace.txt): epilogue 11 k cycles,
  // staging + bookkeeping 13 k per tile next to 31 k of MFMAs; with this structure 1.8-2.2 k (LDS stores alone).  Inside the
  // k-steps the same instructions issue in the shadow of the wave's own 32-cycle MFMAs.  Effect on the kernel: MFMA pipe busy
  // 90.2 -> 92.3 % of the SIMD cycles, 4.85 -> 4.83 ms (the copy costs 1 % more MFMAs).
  f32x4 sav[MT][NT];
  float2 sav_ac[NT];
  Tile prev{};
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) sav[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < NT; ++n) sav_ac[n] = float2{0.f, 0.f};
  // (compile-time n, wv: as run-time lambda arguments they turned `sav` into a scratch array)
  auto epi_part = [&](auto N_, auto WV_) __attribute__((always_inline)) {   // the pooled output(s) of window column wv, channel tile n, of `prev`
    constexpr int n = decltype(N_)::value, wv = decltype(WV_)::value;
    const int j = lane & 15, g = lane >> 4;
    const int ch = n * 16 + j;
    float* orow = p.out + ((size_t)prev.clip * p.nsub + prev.band) * C::COUT * p.out_rows * p.out_cols + (size_t)prev.tr * p.out_cols;
    const float2 ac = sav_ac[n];
    const int pc = 4 * C::WPG * prev.tc + C::WPG * g + wv;
    if (SUB == 2 && p.pool_h == 1) {   // two 1 x 5 windows: positions 0..4 are tile row 0, 5..9 tile row 1
#pragma unroll
      for (int hr = 0; hr < 2; ++hr) {
        float m = 0.f;
#pragma unroll
        for (int pos = 0; pos < 5; ++pos) {
          const int e = wv * C::WIN + 5 * hr + pos;
          m = fmaxf(m, fmaf(sav[e >> 2][n][e & 3], ac.x, ac.y));
        }
        if (prev.valid && pc < p.out_cols && 2 * prev.tr + hr < p.out_rows)
          orow[(size_t)ch * p.out_rows * p.out_cols + (size_t)(prev.tr + hr) * p.out_cols + pc] = m;   // orow is at row tr: + tr + hr
      }
    } else {
      float m = 0.f;
#pragma unroll
      for (int pos = 0; pos < C::WIN; ++pos) {
        const int e = wv * C::WIN + pos;
        m = fmaxf(m, fmaf(sav[e >> 2][n][e & 3], ac.x, ac.y));
      }
      if (prev.valid && pc < p.out_cols) orow[(size_t)ch * p.out_rows * p.out_cols + pc] = m;
    }
  };
  auto tile_dyn = [&]() __attribute__((always_inline)) {
#ifdef MST_TRACE
    const long long tr0 = clock64();
#endif
    float2 e_ac[NT];   // requested now, used by the deferred epilogue
    {
      const float2* aff = p.aff + ((size_t)cur.clip * p.nsub + cur.band) * C::COUT;
#pragma unroll
      for (int n = 0; n < NT; ++n) e_ac[n] = aff[n * 16 + (lane & 15)];
    }
    {
      constexpr int kEpi0 = 12;   // first k-step that carries a part of the previous tile's epilogue (one part every second step)
      // The 98 k-steps in three blocks of tap rows: A = rows 2, 3, 4 (42 steps, every tile), B = rows 5, 6, C = rows 0, 1 (28 each).
      // The FIRST tile row of a band sees nothing but zero padding through tap rows 0 and 1 (input rows -3 .. -1), the LAST one through
      // rows 5 and 6: those tiles skip block C / block B -- 28 of 98 steps for 2 of every 10 tile rows, 5.7 % of the kernel's MFMAs.
      // (Adding the skipped products would add exact zeros: the weights are finite.)  All side work rides in block A.
      const bool skip_c = cur.tr == 0, skip_b = 2 * cur.tr + 2 >= p.in_rows;
      const int next_off = skip_b ? 0 : 5 * PC, next_w = skip_b ? 0 : 35 * NT * 64;   // first step after block A: of C or of B
      float a[2][MT], b[2][NT];
#pragma unroll
      for (int t = 0; t < MT; ++t) a[0][t] = pbuf[abase[t] + 2 * PC];   // step 0: tap row 2, column 0, channels 0..3
#pragma unroll
      for (int n = 0; n < NT; ++n) b[0][n] = wres[(14 * NT + n) * 64 + lane];
      // (static_for: with the ticket loop inside, `#pragma unroll` no longer unrolled the steps and every compile-time index
      // below became a run-time one)
      auto kstep = [&](auto S_) __attribute__((always_inline)) {
        constexpr int ks = decltype(S_)::value;                                 // position in the sequence A, B, C
        constexpr int cu = ks & 1, nx = cu ^ 1;                                 // (every block has an even number of steps)
        constexpr bool last = ks == 41 || ks == 69 || ks == 97;                 // last step of its block
        constexpr int s1 = ks + 1, l1 = s1 < 42 ? s1 : s1 < 70 ? s1 - 42 : s1 - 70;
        constexpr int row1 = (s1 < 42 ? 2 : s1 < 70 ? 5 : 0) + l1 / 14, tap1 = 7 * row1 + (l1 % 14) / 2, ch1 = l1 & 1;
        constexpr int off1 = ch1 * 4 * CHS + row1 * PC + (l1 % 14) / 2;
#pragma unroll
        for (int i = 0; i < MT * NT; ++i) {
          const int t = i / NT, n = i % NT;
          // (step 0 starts from a literal zero: no zeroing pass over the accumulators)
          acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cu][t], b[cu][n], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[t][n], 0, 0, 0);
          if constexpr (!last) {          // fragments of the next step of this block
            if (i < NT) b[nx][i] = wres[ch1 * WCH + (tap1 * NT + i) * 64 + lane];
            else if (i < MT + NT) a[nx][i - NT] = pbuf[abase[i - NT] + off1];
          } else if constexpr (ks == 41) {   // ... of the first step of B or C (whichever this tile runs next)
            if (i < NT) b[nx][i] = wres[next_w + i * 64 + lane];
            else if (i < MT + NT) a[nx][i - NT] = pbuf[abase[i - NT] + next_off];
          } else if constexpr (ks == 69) {   // ... of the first step of C
            if (i < NT) b[nx][i] = wres[i * 64 + lane];
            else if (i < MT + NT) a[nx][i - NT] = pbuf[abase[i - NT]];
          }
          if (i == MT + NT) {   // the next tile's patch, fetched in block A (its set-up is in step 6)
            if constexpr (LAY == 0) {
              if constexpr (ks >= 8 && ks < 8 + NPF / 2) prefetch_piece(2 * (ks - 8)), prefetch_piece(2 * (ks - 8) + 1);
            } else {
              if constexpr (ks >= 8 && ks % 2 == 0 && ks / 2 - 4 < NPF) prefetch_piece(ks / 2 - 4);
            }
          }
          if (i == MT + NT + 1) {
            if constexpr (ks == 0) ticket_issue();
            if constexpr (ks == 6) {
              nxt = ticket_take();
              nxt_seg = dseg;
              prefetch_setup(nxt);
            }
            if constexpr (ks >= kEpi0 && ks < kEpi0 + 2 * NT * C::WPG && (ks - kEpi0) % 2 == 0)
              epi_part(std::integral_constant<int, (ks - kEpi0) / 2 / C::WPG>{}, std::integral_constant<int, (ks - kEpi0) / 2 % C::WPG>{});
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      static_assert(LAY == 0 ? 8 + NPF / 2 <= 42 : 2 * (NPF + 3) < 42, "the patch fetch must fit block A");
      static_for<42>([&](auto I) __attribute__((always_inline)) { kstep(std::integral_constant<int, decltype(I)::value>{}); });
      if (!skip_b) static_for<28>([&](auto I) __attribute__((always_inline)) { kstep(std::integral_constant<int, 42 + decltype(I)::value>{}); });
      if (!skip_c) static_for<28>([&](auto I) __attribute__((always_inline)) { kstep(std::integral_constant<int, 70 + decltype(I)::value>{}); });
    }
    // hand the tile to the deferred epilogue.  The copy is made BY THE MATRIX PIPE (D = 0 x B + C, exact: the weights are finite;
    // 10 MFMAs = 1 % of the tile's): as 40 v_mov after the last k-step it would be VALU work of a wave outside its k-steps
    {
      const float bz = wres[lane];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) sav[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(0.f, bz, acc[t][n], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) sav_ac[n] = e_ac[n];
    prev = cur;
#ifdef MST_TRACE
    trc[0] += clock64() - tr0, trc[4] += 1;
#endif
  };
  auto load_band = [&](int band) __attribute__((always_inline)) {
    for (int c = 0; c < 2; ++c) {
      const f32x4* src = reinterpret_cast<const f32x4*>(p.wfrag + ((size_t)band * 2 + c) * WBP);
      for (int k = tid; k < WCH / 4; k += kConvThreads) reinterpret_cast<f32x4*>(wres + c * WCH)[k] = src[k];
    }
  };
  if constexpr (DYN) {
    // Tiles by ticket.  The two waves that share a SIMD do not advance at the same rate (traced, MST_TRACE build: the waves
    // dispatched first take ~57 % of a SIMD's MFMA slots), so with tiles dealt statically (wave w takes tile 8 s + w) the faster
    // waves finished early and their partners ran the rest alone.  Here a wave takes the next tile of the workgroup's run when
    // it is ready for one, and both waves of a SIMD work until the run is empty.  Results do not depend on which wave computes a
    // tile.  (A start-up offset between the two waves of a SIMD -- s_sleep in one of them -- changed nothing.)
    for (int k = 0; k < nseg; ++k) {
#ifdef MST_TRACE
      const long long tb0 = clock64();
#endif
      __syncthreads();                                   // every wave is done with the previous segment's weights
      load_band((gb0 + k) % p.nsub);
      if (tid == 0) tickets[(k + kTickets - 1) % kTickets] = 0;   // segment k - 1's counter: next used by segment k + kTickets - 1
      __syncthreads();
#ifdef MST_TRACE
      trc[3] += clock64() - tb0;
#endif
      while (nxt.valid && nxt_seg == k) {
#ifdef MST_TRACE
        const long long ts0 = clock64();
#endif
        cur = nxt;
#ifdef MST_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long ts1 = clock64();
        trc[1] += ts1 - ts0;
#endif
        stage_(std::true_type{});   // LDS stores only: VALU instructions of a wave outside its k-steps wait for the partner's MFMA stream
#ifdef MST_TRACE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        trc[2] += clock64() - ts1;
#endif
        tile_dyn();
      }
    }
    {   // the last tile's epilogue
      static_for<NT * C::WPG>([&](auto I) __attribute__((always_inline)) {
        constexpr int q = decltype(I)::value;
        epi_part(std::integral_constant<int, q / C::WPG>{}, std::integral_constant<int, q % C::WPG>{});
      });
    }
#ifdef MST_TRACE
    if (lane == 0 && p.yraw) {
      float* o = p.yraw + (blockIdx.x * kConvWaves + wave) * 8;
      for (int i = 0; i < 5; ++i) o[i] = (float)trc[i];
      o[5] = (float)(clock64() - trc_start);
    }
#endif
  } else {
  // (staging the NEXT tile's patch before this tile's epilogue -- so that no tile opens behind the acknowledgement of the previous
  // tile's stores in the in-order vmcnt counter -- measured 0.7 % SLOWER here: the 31 k MFMA cycles of a tile dwarf that wait)
  for (int s = s_begin; s < s_end; ++s) {
    cur = nxt;
    if (cur.band != cur_band) {  // same `s` sequence in every wave: all eight reach this together
      if (MODE == 1 && cur_band >= 0) flush_stats<NT, C::COUT>(st, p.stats, cur_band, lane);
      __syncthreads();
      load_band(cur.band);
      __syncthreads();
      cur_band = cur.band;
    }
    stage();
    nxt = decode(s + 1);
    prefetch_setup(nxt);
    if (!cur.valid) {
#pragma unroll
      for (int i = 0; i < NPF; ++i) prefetch_piece(i);
      continue;
    }
    tile_body();
  }
  }
  if (MODE == 1 && cur_band >= 0) flush_stats<NT, C::COUT>(st, p.stats, cur_band, lane);
}

// ------------------------------------------------------------------------------------------
// OPT-IN conv1 on the f16 matrix cores with 3-term split precision (mst_encoder_set_precision(enc, 1)).
// x = xh + xl, w = wh + wl (f16 each, 22 significant bits together; weights pre-scaled by 2^10 so that wl stays a
// normal f16), x*w ~= xh*wh + xh*wl + xl*wh exactly representable products, fp32 accumulation in the MFMA
// (v_mfma_f32_16x16x32_f16).  Per-product relative error ~2^-22 (fp32: 2^-24); parity-tested at the same 1e-4 bar.
// Same tile / epilogue structure as conv1_resident_kernel; a k-step is 4 taps x 8 input channels (13 steps, the
// last 3 tap slots carry zero weights); the patch is stored channel-minor ([row][col][8 ch] f16, hi and lo) so an
// A fragment is one ds_read_b128.  The default path stays the exact-fp32 kernel above.
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Range control of the f16 split-precision path.  conv2 reads conv1's pooled activations as f16 (hi + lo); f16
// saturates at 65504 and loses its low part below ~1e-1.  Instead of checking the range on the host, every
// (clip, band) gets an exact power-of-two scale s from a rigorous bound, before conv1 runs:
//     pooled = max(0, A_c * acc + C_c),  |acc| <= ||w_c||_1 * max|log-mel of the clip|
//     bound  = max_c (|A_c| ||w_c||_1 xmax + |C_c|),   s = 2^(14 - ceil(log2 bound))   =>   s * pooled <= 2^14.
// The bound is loose (typically 30-100x), which f16's floating exponent absorbs: values 2^-28 of the bound are still
// normal.  s is exact in every format involved, so scaling changes no rounding except through the f16 range itself.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n_per_clip, unsigned* out) {
  const float* base = x + (size_t)blockIdx.y * n_per_clip;
  float m = 0.f;
  const long long n4 = n_per_clip >> 2;
  const float4* b4 = reinterpret_cast<const float4*>(base);
  const bool al = (reinterpret_cast<uintptr_t>(base) & 15) == 0;
  if (al) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
      const float4 v = b4[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n_per_clip; i += (long long)gridDim.x * 256)
      m = fmaxf(m, fabsf(base[i]));
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_per_clip; i += (long long)gridDim.x * 256)
      m = fmaxf(m, fabsf(base[i]));
  }
  m = mst::wave_max(m);
  // non-negative floats order like their bit patterns; a NaN / Inf input ends up as a huge pattern -> scale 2^-60
  if ((threadIdx.x & 63) == 0) atomicMax(out + blockIdx.y, __float_as_uint(m));
}

// bias / gain: the training forward's affine acts on (acc + bias) and Dropout rescales the pooled value by `gain`
__global__ __launch_bounds__(64) void f16_scale_kernel(const float2* __restrict__ aff1, const float* __restrict__ w1norm,
                                                        const unsigned* __restrict__ xmax_bits, float* scale, int nsub,
                                                        const float* __restrict__ bias = nullptr, float gain = 1.0f) {
  const int clip = blockIdx.x / nsub, band = blockIdx.x % nsub, c = threadIdx.x & 31;
  const float xmax = __uint_as_float(xmax_bits[clip]);
  const float2 a = aff1[((size_t)clip * nsub + band) * 32 + c];
  float bound = fmaf(fabsf(a.x), fmaf(w1norm[band * 32 + c], xmax, bias ? fabsf(bias[band * 32 + c]) : 0.f), fabsf(a.y)) * gain;
  bound = mst::wave_max(bound);
  if (threadIdx.x == 0) {
    int k = 0;
    if (bound > 0.f && bound < INFINITY) {
      int ex;
      (void)frexpf(bound, &ex);          // bound = f * 2^ex, f in [0.5, 1)  =>  bound <= 2^ex
      k = max(-60, min(60, 14 - ex));
    } else if (!(bound < INFINITY)) {
      k = -60;
    }
    scale[(size_t)blockIdx.x * 2] = ldexpf(1.0f, k);
    scale[(size_t)blockIdx.x * 2 + 1] = ldexpf(1.0f, -k);
  }
}

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// saturating conversion of FINITE values (the power-of-two range scales keep real data far below the ceiling; this only guards the
// bound's slack); a NaN or an infinity stays one -- fminf / fmaxf would turn a NaN into -65504 and hide a diverged step from the
// caller's GradScaler / NaN checks (src/train.py:251-262 skips such steps)
__device__ __forceinline__ _Float16 to_f16_sat(float v) {
  return (_Float16)(__builtin_isfinite(v) ? fminf(fmaxf(v, -65504.f), 65504.f) : v);
}
// weight pre-scale of the f16 fragments: an exact power of two per (band, output channel), chosen on the host so that the
// filter's largest |w| lands in [2^13, 2^14) -- wl = w - wh stays a normal f16 whatever the weights' magnitude -- and
// folded back per channel in the epilogue (ConvParams::f16_winv)
__host__ inline int f16_weight_exponent(float maxabs) {
  if (!(maxabs > 0.f) || !(maxabs < INFINITY)) return 10;
  int ex;
  (void)frexpf(maxabs, &ex);   // maxabs in [2^(ex-1), 2^ex)
  return std::max(-100, std::min(100, 14 - ex));
}
constexpr int kF16Steps = 13;         // ceil(49 taps / 4)

// TERMS = 3: split precision (fp32-equivalent, see above).  TERMS = 1: plain f16 operands (`x ~ xh`, `w ~ wh`), fp32
// accumulate -- the arithmetic of the reference's `--use_amp` autocast convolutions (src/train.py:251-253), opt-in.
// MODE 1 (training forward, TERMS = 1): raw convolution output + bias in accumulator order and the batch-statistics sums,
// exactly as conv1_resident_kernel<SUB, 1> leaves them.
// LAY 0: fp32 log-mel in the reference layout, converted to float16 hi / lo while it is staged (64 row loads and ~700 vector
//        instructions per tile: more time than the MFMAs of the plain-f16 mode).
// LAY 2: the log-mel arrives from stage A as float16 hi / lo planes, channel-minor [B][F][M][8] (MST_LOGMEL_CM16): a patch
//        position IS one 16-byte vector of the LDS image -- 6 (+ 6 low-part) loads and as many ds_write_b128 per tile, no
//        conversion, the same bits as LAY 0 produces.
template <int SUB, int TERMS, int MODE = 0, int LAY = 0>
__global__ __launch_bounds__(kConvThreads) void conv1_f16x3_kernel(const ConvParams p, const h16x8* __restrict__ wfrag16,
                                                                  _Float16* __restrict__ out_hi, _Float16* __restrict__ out_lo) {
  using C = CC<1, SUB>;
  constexpr int MT = C::MT, NT = C::NT, PR = C::PR, PC = C::PC;
  static_assert(C::RL == 64, "one loader row per instruction");
  static_assert(LAY == 0 || (LAY == 2 && PR == 8), "the channel-minor loader maps lane bits 0..2 to the 8 patch rows");
  constexpr int WVEC = kF16Steps * NT * 2 * 64;      // h16x8 vectors of one band's weights: [step][nt][hi/lo][lane]
  constexpr int PVEC = PR * PC;                      // positions per patch (one h16x8 = 8 channels each)
  constexpr int NPF = LAY == 0 ? 8 * PR : (PC + 7) / 8;   // LAY 2: 8 frames x 8 rows per load instruction
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  h16x8* wres = reinterpret_cast<h16x8*>(smem);                  // [WVEC]
  h16x8* phi = wres + WVEC + wave * 2 * PVEC;                     // wave-private patch, hi part
  h16x8* plo = phi + PVEC;                                        //                     lo part

  const int G = gridDim.x;
  const int wg = mst::xcd_remap(blockIdx.x, G);
  // Order of the sets.  Band-major (clip_major = 0; the training forward: the batch statistics are flushed at every band change,
  // so a workgroup should see as few as possible): sets_per_band sets per band, the clips' tiles concatenated inside it.
  // Clip-group-major (clip_major = k > 0; the eval forward): the clips are taken k at a time, and inside a group the bands follow
  // each other ([group][band][clip of the group][tile]; sets_per_band counts the sets of ONE (group, band)).  Measured on 72
  // clips (FETCH_SIZE x 2 per launch, kernel time equal within +-0.5 % for all of them): band-major 1.93 GB, k = 4 1.77 GB,
  // k = 1 1.11 GB = 72 x 11 patches of 26 mel rows, i.e. every (clip, band) patch fetched exactly once; the sub-bands' shared
  // rows (each mel row sits in two bands: 0.51 GB algorithmic) are never in L2 long enough in any order -- a workgroup meets them
  // again a whole (clip, band) later.  k = 1 pads the 430 tiles of a (clip, band) to 54 sets (0.5 % empty wave slots).
  const int total_sets = (p.clip_major ? (p.B + p.clip_major - 1) / p.clip_major : 1) * p.nsub * p.sets_per_band;
  const int s_begin = (int)((long long)wg * total_sets / G), s_end = (int)((long long)(wg + 1) * total_sets / G);
  const int tpb = p.tiles_r * p.tiles_c;

  auto decode = [&](int s) __attribute__((always_inline)) {
    Tile t;
    int ti;
    if (p.clip_major) {   // groups of clip_major clips: [group][band][clip of the group][tile]
      const int ngrp = (p.B + p.clip_major - 1) / p.clip_major;
      const int gb = min(s / p.sets_per_band, ngrp * p.nsub - 1);
      const int idx = (s - gb * p.sets_per_band) * kConvWaves + wave;
      const int grp = gb / p.nsub, c0 = grp * p.clip_major;
      t.band = gb - grp * p.nsub;
      t.valid = (s < s_end) && idx < min(p.clip_major, p.B - c0) * tpb;
      const int cl = t.valid ? idx / tpb : 0;
      t.clip = c0 + cl;
      ti = t.valid ? idx - cl * tpb : 0;
    } else {
      t.band = min(s / p.sets_per_band, p.nsub - 1);
      const int idx = (s - t.band * p.sets_per_band) * kConvWaves + wave;
      t.valid = (s < s_end) && idx < p.B * tpb;
      t.clip = t.valid ? idx / tpb : 0;
      ti = t.valid ? idx - t.clip * tpb : 0;
    }
    t.tc = ti / p.tiles_r;
    t.tr = ti - t.tc * p.tiles_r;
    return t;
  };

  const int kq = lane >> 4, ai = lane & 15, ag = ai >> 2, areg = ai & 3;
  int abase[MT];   // position index of A row (lane & 15) of M-tile t inside the patch
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int e = 4 * t + areg, wv = e / C::WIN, pos = e % C::WIN;
    abase[t] = (pos / 5) * PC + 5 * (C::WPG * ag + wv) + pos % 5;
  }
  int toff[kF16Steps];  // patch offset of this lane group's tap in every k-step (tap = 4 s + kq; padded slots reuse tap 48)
#pragma unroll
  for (int st = 0; st < kF16Steps; ++st) {
    const int tap = min(4 * st + kq, 48);
    toff[st] = (tap / 7) * PC + tap % 7;
  }

  typename std::conditional<LAY == 0, float, h16x8>::type pf[NPF];
  h16x8 pfl[LAY == 0 ? 1 : NPF];     // LAY 2, TERMS 3: the low parts
  unsigned long long rowmask = 0;    // LAY 0: bit i = row of piece i inside the band (wave-uniform); LAY 2: bit i = this lane's position of piece i inside
  bool col_ok = false;               // LAY 0: this lane's column inside the clip; LAY 2: this lane's row inside the band
  int coff = 0, nrow0 = 0, nvalid = 0;
  const float* nsrc = p.in;
  const char* nsrc_lo = static_cast<const char*>(p.in_lo);
  // LAY 2: lane = 8 fq + r -> patch row r, frames fq + 8 i (piece i)
  const int lr = lane & 7, lfq = lane >> 3;
  const unsigned frame_bytes = (unsigned)p.cm_mels * 16u;
  auto prefetch_setup = [&](const Tile& t) __attribute__((always_inline)) {
    nrow0 = C::TROWS * t.tr - 3;
    nvalid = t.valid;
    rowmask = 0;
    if constexpr (LAY == 0) {
      const int cin = C::TCOLS * t.tc - 3 + lane;
      nsrc = p.in + (size_t)t.clip * p.in_clipstride + (size_t)t.band * p.in_bandoff;
      col_ok = lane < PC && cin >= 0 && cin < p.in_cols;
      coff = min(max(cin, 0), p.in_cols - 1);
    } else {
      const int rin = nrow0 + lr, rc = min(max(rin, 0), p.in_rows - 1);
      const size_t cb = (size_t)t.clip * p.in_clipstride * 2;               // bytes: in_clipstride counts float16 elements
      nsrc = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.in) + cb);   // wave-uniform clip base (high parts)
      nsrc_lo = static_cast<const char*>(p.in_lo) + cb;
      col_ok = nvalid && rin == rc;
      coff = (t.band * p.cm_overlap + rc) * 16;                              // byte offset of this lane's row inside a frame
      nrow0 = C::TCOLS * t.tc - 3 + lfq;                                     // (re-used: first frame of this lane)
    }
  };
  auto prefetch_piece = [&](int i) __attribute__((always_inline)) {
    if (i < NPF) {
      if constexpr (LAY == 0) {
        const int cc = i / PR, r = i % PR;
        const int rin = nrow0 + r;
        const int rc = min(max(rin, 0), p.in_rows - 1);
        if (nvalid && rin == rc) rowmask |= 1ull << i;
        pf[i] = (nsrc + (size_t)cc * p.in_cstride + (size_t)rc * p.in_cols)[coff];
      } else {
        const int fin = nrow0 + 8 * i, fc = min(max(fin, 0), p.in_cols - 1);
        if (col_ok && fin == fc) rowmask |= 1ull << i;
        const unsigned boff = (unsigned)fc * frame_bytes + (unsigned)coff;   // 32-bit offset from the scalar clip base
        pf[i] = *reinterpret_cast<const h16x8*>(reinterpret_cast<const char*>(nsrc) + boff);
        if (TERMS == 3) pfl[i] = *reinterpret_cast<const h16x8*>(nsrc_lo + boff);
      }
    }
  };

  f32x4 acc[MT][NT];
  double st[NT][2];
#pragma unroll
  for (int n = 0; n < NT; ++n) st[n][0] = st[n][1] = 0.0;
  int cur_band = -1;
  Tile cur{}, nxt = decode(s_begin);
  prefetch_setup(nxt);
#pragma unroll
  for (int i = 0; i < NPF; ++i) prefetch_piece(i);

  for (int s = s_begin; s < s_end; ++s) {
    cur = nxt;
    if (cur.band != cur_band) {
      if (MODE >= 1 && cur_band >= 0) flush_stats<NT, C::COUT>(st, p.stats, cur_band, lane);
      __syncthreads();
      const h16x8* src = wfrag16 + (size_t)cur.band * WVEC;
      for (int k = tid; k < WVEC; k += kConvThreads) wres[k] = src[k];
      __syncthreads();
      cur_band = cur.band;
    }
    // stage the prefetched patch as f16 hi / lo, channel-minor: one 16-byte vector per position
    if constexpr (LAY == 2) {
      const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < NPF; ++i) {
        if (8 * i + 7 < PC || 8 * i + lfq < PC) {
          const bool ok = (rowmask >> i) & 1ull;
          phi[lr * PC + lfq + 8 * i] = ok ? pf[i] : z;
          if (TERMS == 3) plo[lr * PC + lfq + 8 * i] = ok ? pfl[i] : z;
        }
      }
    } else if (lane < PC) {
#pragma unroll
      for (int r = 0; r < PR; ++r) {
        h16x8 vh, vl;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const int i = cc * PR + r;
          const float x = (((rowmask >> i) & 1ull) && col_ok) ? pf[i] : 0.f;
          const _Float16 h = (_Float16)x;
          vh[cc] = h;
          vl[cc] = (_Float16)(x - (float)h);
        }
        phi[r * PC + lane] = vh;
        if (TERMS == 3) plo[r * PC + lane] = vl;
      }
    }
    nxt = decode(s + 1);
    prefetch_setup(nxt);
    if (!cur.valid) {
#pragma unroll
      for (int i = 0; i < NPF; ++i) prefetch_piece(i);
      continue;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // fragments of k-step st+1 are read from LDS while the 30 MFMAs of step st run (double-buffered registers)
      h16x8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bh[0][n] = wres[(n * 2 + 0) * 64 + lane];
        if (TERMS == 3) bl[0][n] = wres[(n * 2 + 1) * 64 + lane];
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        ah[0][t] = phi[abase[t] + toff[0]];
        if (TERMS == 3) al[0][t] = plo[abase[t] + toff[0]];
      }
#pragma unroll
      for (int st = 0; st < kF16Steps; ++st) {
        const int cu = st & 1, nx = cu ^ 1;
        if (st + 1 < kF16Steps) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            bh[nx][n] = wres[(((st + 1) * NT + n) * 2 + 0) * 64 + lane];
            if (TERMS == 3) bl[nx][n] = wres[(((st + 1) * NT + n) * 2 + 1) * 64 + lane];
          }
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            ah[nx][t] = phi[abase[t] + toff[st + 1]];
            if (TERMS == 3) al[nx][t] = plo[abase[t] + toff[st + 1]];
          }
        }
        if constexpr (LAY == 0) {
#pragma unroll
          for (int k = 0; k < 5; ++k) prefetch_piece(st * 5 + k);   // 64 row loads of the next tile over 13 steps
        } else {
          if (st % 2 == 0) prefetch_piece(st / 2);                  // 6 (+ 6) 16-byte loads over 13 steps
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if (TERMS == 3) {
              acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cu][t], bh[cu][n], acc[t][n], 0, 0, 0);  // small terms first
              acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cu][t], bl[cu][n], acc[t][n], 0, 0, 0);
            }
            acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cu][t], bh[cu][n], acc[t][n], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);   // keep the next step's reads ahead of this step's MFMAs, per step
      }
    }
    if constexpr (MODE >= 1) {
      const int j = lane & 15, g = lane >> 4;
      const size_t unit0 = ((((size_t)cur.clip * p.nsub + cur.band) * p.acc_tr + cur.tr) * p.acc_tc + cur.tc) * (size_t)(NT * 64);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float b = p.bias[cur.band * C::COUT + n * 16 + j];
        const float wi = p.f16_winv[cur.band * C::COUT + n * 16 + j];   // undo the weight pre-scale (exact power of two)
        const size_t unit = unit0 + (size_t)(n * 64 + lane);
        const float sy = MODE == 2 ? p.y_scale[cur.band * 2] : 1.f, isy = MODE == 2 ? p.y_scale[cur.band * 2 + 1] : 1.f;
        float ps = 0.f, pq = 0.f;   // the tile's 20 values in fp32, then one fold into the double accumulators (see conv2_f16x3_kernel)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          f32x4 v = acc[t][n];
          h16x4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = fmaf(v[r], wi, b);
            if constexpr (MODE == 2) {   // the value that is stored, and that the statistics are taken of
              h[r] = to_f16_sat(v[r] * sy);
              v[r] = (float)h[r] * isy;
            }
            const int e = 4 * t + r, wv = e / C::WIN, pos = e % C::WIN;
            const int col = C::TCOLS * cur.tc + 5 * (C::WPG * g + wv) + pos % 5;
            if (col < p.raw_cols) ps += v[r], pq = fmaf(v[r], v[r], pq);
          }
          if constexpr (MODE == 2) reinterpret_cast<h16x4*>(p.yraw16)[yvec(unit, t, MT)] = h;   // t-major: whole-line stores
          else reinterpret_cast<f32x4*>(p.yraw)[yvec(unit, t, MT)] = v;
        }
        st[n][0] += (double)ps, st[n][1] += (double)pq;
      }
    } else {
      const int j = lane & 15, g = lane >> 4;
      const float2* aff = p.aff + ((size_t)cur.clip * p.nsub + cur.band) * C::COUT;
      float* orow = p.out + ((size_t)cur.clip * p.nsub + cur.band) * C::COUT * p.out_rows * p.out_cols +
                    (size_t)cur.tr * p.out_cols;
      const float f16s = p.f16_scale ? p.f16_scale[((size_t)cur.clip * p.nsub + cur.band) * 2] : 1.0f;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int ch = n * 16 + j;
        float2 ac = aff[ch];
        ac.x *= p.f16_winv[cur.band * C::COUT + ch];  // undo the weight pre-scale
#pragma unroll
        for (int wv = 0; wv < C::WPG; ++wv) {
          const int pc = 4 * C::WPG * cur.tc + C::WPG * g + wv;
          const int nh = (SUB == 2 && p.pool_h == 1) ? 2 : 1;   // pool height 1: two 1 x 5 windows (tile rows 0 and 1) per lane
#pragma unroll
          for (int hr = 0; hr < 2; ++hr) {
            if (hr < nh) {
              float m = 0.f;
#pragma unroll
              for (int pos = 0; pos < C::WIN; ++pos) {
                const int e = wv * C::WIN + pos;
                const bool mine = nh == 1 || pos / 5 == hr;
                m = mine ? fmaxf(m, fmaf(acc[e >> 2][n][e & 3], ac.x, ac.y)) : m;
              }
              const int orow_i = nh == 1 ? cur.tr : 2 * cur.tr + hr;   // output row
              if (pc < p.out_cols && orow_i < p.out_rows) {
                if (p.out) orow[(size_t)ch * p.out_rows * p.out_cols + (size_t)(orow_i - cur.tr) * p.out_cols + pc] = m;
                if (out_hi) {  // conv2's split-precision input: [clip][band][row][col][32 ch] f16, hi and lo, range-scaled
                  const size_t o = ((((size_t)cur.clip * p.nsub + cur.band) * p.out_rows + orow_i) * p.out_cols + pc) * 32 + ch;
                  const float ms = m * f16s;
                  const _Float16 h = (_Float16)ms;
                  out_hi[o] = h;
                  if (TERMS == 3) out_lo[o] = (_Float16)(ms - (float)h);
                }
              }
            }
          }
        }
      }
    }
  }
  if (MODE >= 1 && cur_band >= 0) flush_stats<NT, C::COUT>(st, p.stats, cur_band, lane);
}


// ------------------------------------------------------------------------------------------
// conv1 on the f16 matrix cores, EVAL path, channel-minor float16 input (MST_LOGMEL_CM16): the bank-conflict-free build.
// conv1_f16x3_kernel's A-fragment reads (one ds_read_b128 per (M-tile, k-step): 16 lanes per LDS cycle, conflict-free only
// if their 16-byte units are distinct mod 16) ran at 8.3 LDS cycles instead of 4 -- SQ_LDS_BANK_CONFLICT 44 % of the LDS
// cycles, LDS busier than the matrix pipe in the plain-f16 mode.  Three changes make every read conflict-free:
//   * A row (lane & 15) = 4 g + 2 dr + dc of M-tile t  <->  position (row dr, column 10 g + 2 t + dc) of the 2 x 40 tile: the
//     four rows of a lane group are a 2 x 2 block, M-tile t walks the group's ten columns in pairs -- every 2 x 5 pooling
//     window still sits in ONE lane's accumulators (tile t = 2 holds the last column of window 0 and the first of window 1);
//   * the patch keeps its even rows at units [0, 4 RS) and its odd rows 8 units further on ((4 RS + 8) from the even part,
//     RS = 48): a 2 x 2 block touches units {c, c + 1, c + 8, c + 9} mod 16, and the four lane groups (columns 10 g) then fill
//     all 16 residues: {0, 10, 4, 14} + {0, 1, 8, 9} = Z_16;
//   * a k-step is ONE tap column and four tap rows, dealt to the k-groups (lane >> 4) as rows (0, 2, 1, 3): the two k-groups
//     that share an LDS cycle differ by two rows = 0 mod 16 units.  7 tap columns x (rows 0..3, rows 4..6 + one zero slot)
//     = 14 k-steps (13 in the other kernel: + 7.7 % MFMAs, all addresses are lane base + immediate).
// Same epilogue arithmetic; the fp32 sums run over the taps in a different order than conv1_f16x3_kernel's.
// ------------------------------------------------------------------------------------------
constexpr int kF16StepsE = 14;
constexpr int kF16eRS = 48, kF16eOdd = 4 * kF16eRS + 8, kF16ePatch = kF16eOdd + 4 * kF16eRS;   // units (16 bytes) per patch part

// Waves per workgroup: 8 with split precision (hi + lo patches: 152 KB of LDS); 12 with plain f16 (104 KB, 3 waves per SIMD on
// <= 168 registers: the third wave covers what two leave open -- per tile a wave has 2240 cycles of MFMAs and about as many of
// staging, epilogue and tile bookkeeping).
template <int TERMS> constexpr int kF16eWaves = TERMS == 1 ? 12 : 8;
template <int TERMS>
__global__ __launch_bounds__(kF16eWaves<TERMS> * 64) void conv1_f16e_kernel(const ConvParams p, const h16x8* __restrict__ wfrag16,
                                                                 _Float16* __restrict__ out_hi, _Float16* __restrict__ out_lo) {
  using C = CC<1, 2>;
  constexpr int MT = C::MT, NT = C::NT, PR = C::PR, PC = C::PC;
  static_assert(PR == 8 && PC <= kF16eRS && MT == 5, "2 x 40 tiles with halo 3");
  constexpr int HL = TERMS == 3 ? 2 : 1, NW = kF16eWaves<TERMS>, NTHR = NW * 64;
  constexpr int WVG = kF16StepsE * NT * 2 * 64;       // h16x8 vectors of one band's weights in global memory: [step][nt][hi/lo][lane]
  constexpr int WV = kF16StepsE * NT * HL * 64;       // ... staged into LDS (plain f16: the high parts only)
  constexpr int NPF = (PC + 7) / 8;                   // 8 rows x 8 frames per load instruction
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  h16x8* wres = reinterpret_cast<h16x8*>(smem);                       // [WV]
  h16x8* phi = wres + WV + wave * HL * kF16ePatch;                     // wave-private patch, hi part
  h16x8* plo = phi + kF16ePatch;                                       //                     lo part (TERMS 3)

  const int G = gridDim.x;
  const int wg = mst::xcd_remap(blockIdx.x, G);
  const int total_sets = p.nsub * p.sets_per_band;
  const int s_begin = (int)((long long)wg * total_sets / G), s_end = (int)((long long)(wg + 1) * total_sets / G);
  const int tpb = p.tiles_r * p.tiles_c;
  auto decode = [&](int s) __attribute__((always_inline)) {   // with integer divisions: at the start and at band changes only
    Tile t;
    t.band = min(s / p.sets_per_band, p.nsub - 1);
    const int idx = (s - t.band * p.sets_per_band) * NW + wave;
    t.valid = (s < s_end) && idx < p.B * tpb;
    t.clip = idx / tpb;
    const int ti = idx - t.clip * tpb;
    t.tc = ti / p.tiles_r;
    t.tr = ti - t.tc * p.tiles_r;
    if (!t.valid) t.clip = t.tc = t.tr = 0;
    return t;
  };
  // the wave's next tile is 8 tiles further on in (clip, tile column, tile row) order: carried on the scalar unit without the
  // three divisions of decode() (~190 scalar instructions per tile, a fifth of the plain-f16 tile's MFMA time)
  int sib = 0;   // set index inside the band
  auto advance = [&](const Tile& c, int s_next) __attribute__((always_inline)) {
    Tile t = c;
    if (++sib == p.sets_per_band || s_next >= s_end || !c.valid) {
      t = decode(s_next);
      sib = s_next - t.band * p.sets_per_band;
      return t;
    }
    t.tr += NW;
    while (t.tr >= p.tiles_r) t.tr -= p.tiles_r, ++t.tc;
    while (t.tc >= p.tiles_c) t.tc -= p.tiles_c, ++t.clip;
    t.valid = t.clip < p.B;
    if (!t.valid) t.clip = t.tc = t.tr = 0;
    return t;
  };

  // A-fragment addressing: unit(r, c) = (r >> 1) * RS + (r & 1) * kF16eOdd + c.  Lane: k-group kq -> tap row kqr (+ 4 in odd
  // steps), A row -> (g, dr, dc).  r = dr + kqr + rb with rb in {0, 4} compile-time; the zero slot (tap row 7: odd steps,
  // kq = 3) reads tap row 6 again (its weights are zero; the address must stay inside the patch)
  const int kq = lane >> 4, ai = lane & 15, ag = ai >> 2, adr = (ai >> 1) & 1, adc = ai & 1;
  const int kqr = ((kq & 1) << 1) | (kq >> 1);          // 0, 2, 1, 3
  const int q0 = adr + kqr, q1 = adr + (kq == 3 ? 2 : kqr);
  const int base0 = (q0 >> 1) * kF16eRS + (q0 & 1) * kF16eOdd + 10 * ag + adc;                       // even steps (tap rows 0..3)
  const int base1 = ((q1 >> 1) + 2) * kF16eRS + (q1 & 1) * kF16eOdd + 10 * ag + adc;                 // odd steps (tap rows 4..6, 6)

  h16x8 pf[NPF], pfl[TERMS == 3 ? NPF : 1];
  bool row_ok = false;
  // The pieces are RAW BUFFER LOADS: a lane whose mel row (or, in the first / last tile column, frame) lies outside the image gets an
  // offset beyond the buffer and reads zeros -- no masks when the patch is staged -- and for every other tile column a piece is one
  // scalar add and the load (lane offset once per tile).  Vector instructions inside an MFMA stream are not free
  // (scripts/ubench_mfma.hip modes 6, 7).
  bool nfast = false;
  __amdgpu_buffer_rsrc_t nrs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(static_cast<const void*>(p.in))), 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t nrs_lo = nrs_hi;
  unsigned nvoff = 0, nsoff = 0;
  int coff = 0, frame0 = 0, nvalid = 0;
  const char* nsrc = static_cast<const char*>(static_cast<const void*>(p.in));
  const char* nsrc_lo = static_cast<const char*>(p.in_lo);
  // loader: lane = 8 r + fq -> patch row r, frames fq + 8 i (piece i): a wave instruction reads 8 frames x 128 contiguous bytes
  // and its 8 lanes per LDS write cycle store 8 consecutive units of one row
  const int lr = lane >> 3, lfq = lane & 7;
  const unsigned frame_bytes = (unsigned)p.cm_mels * 16u;
  auto prefetch_setup = [&](const Tile& t) __attribute__((always_inline)) {
    const int rin = C::TROWS * t.tr - 3 + lr, rc = min(max(rin, 0), p.in_rows - 1);
    nvalid = t.valid;
    const size_t cb = (size_t)t.clip * p.in_clipstride * 2;                 // bytes: in_clipstride counts float16 elements
    nsrc = static_cast<const char*>(static_cast<const void*>(p.in)) + cb;   // wave-uniform clip base (high parts)
    nsrc_lo = static_cast<const char*>(p.in_lo) + cb;
    row_ok = nvalid && rin == rc;
    coff = (t.band * p.cm_overlap + rc) * 16;
    frame0 = C::TCOLS * t.tc - 3 + lfq;
    const int f0 = C::TCOLS * t.tc - 3;
    nfast = nvalid && f0 >= 0 && f0 + 8 * NPF <= p.in_cols;
    nrs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(nsrc), 0, (unsigned)p.in_cols * frame_bytes, 0x00020000);
    if (TERMS == 3) nrs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(nsrc_lo), 0, (unsigned)p.in_cols * frame_bytes, 0x00020000);
    nsoff = (unsigned)max(f0, 0) * frame_bytes;
    nvoff = row_ok ? (unsigned)coff + (unsigned)lfq * frame_bytes : 0x80000000u;
  };
  auto prefetch_piece = [&](int i) __attribute__((always_inline)) {
    if (i < NPF) {
      unsigned vo = nvoff, so = nsoff + (unsigned)(8 * i) * frame_bytes;
      if (!nfast) {   // first / last tile column (or no tile): per-lane frame validity
        const int fin = frame0 + 8 * i;
        vo = (row_ok && fin >= 0 && fin < p.in_cols) ? (unsigned)fin * frame_bytes + (unsigned)coff : 0x80000000u;
        so = 0;
      }
      pf[i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(nrs_hi, vo, so, 0));
      if (TERMS == 3) pfl[i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(nrs_lo, vo, so, 0));
    }
  };

  f32x4 acc[MT][NT];
  int cur_band = -1;
  Tile cur{}, nxt = decode(s_begin);
  sib = s_begin - nxt.band * p.sets_per_band;
  prefetch_setup(nxt);
#pragma unroll
  for (int i = 0; i < NPF; ++i) prefetch_piece(i);

  // stage the prefetched patch: one 16-byte vector per position, rows split into the even and the odd part
  auto stage = [&]() __attribute__((always_inline)) {   // LDS stores only (units outside the image were loaded as zeros)
    const int wbase = (lr >> 1) * kF16eRS + (lr & 1) * kF16eOdd + lfq;
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      if (8 * i + 7 < PC || 8 * i + lfq < PC) {
        phi[wbase + 8 * i] = pf[i];
        if (TERMS == 3) plo[wbase + 8 * i] = pfl[i];
      }
    }
  };
  // Order inside a tile: k-steps (the next tile's patch is fetched meanwhile) -> stage the NEXT tile's patch -> this tile's
  // epilogue.  Staging waits for the prefetched loads with a vmcnt wait, and stores count in the same in-order counter: with the
  // epilogue first, every tile opened by sitting out the acknowledgement of the previous tile's output stores.
  stage();
  for (int s = s_begin; s < s_end; ++s) {
    cur = nxt;
    if (cur.band != cur_band) {
      __syncthreads();
      const h16x8* src = wfrag16 + (size_t)cur.band * WVG;
      for (int k = tid; k < WV; k += NTHR) wres[k] = HL == 2 ? src[k] : src[((k >> 6) * 2) * 64 + (k & 63)];
      __syncthreads();
      cur_band = cur.band;
    }
    nxt = advance(cur, s + 1);
    prefetch_setup(nxt);
    if (!cur.valid) {   // ragged tail of a band: nothing to compute, but the next tile's patch must be in place
#pragma unroll
      for (int i = 0; i < NPF; ++i) prefetch_piece(i);
      stage();
      continue;
    }
    // the epilogue's per-(clip, band, channel) constants: requested NOW, used after the k-steps (fetched in the epilogue their
    // global-memory round trips stood in the open once per tile)
    float2 e_ac[NT];
    float e_winv[NT];
    {
      const float2* aff = p.aff + ((size_t)cur.clip * p.nsub + cur.band) * C::COUT;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        e_ac[n] = aff[n * 16 + (lane & 15)];
        e_winv[n] = p.f16_winv[cur.band * C::COUT + n * 16 + (lane & 15)];
      }
    }
    float f16s = p.f16_scale ? p.f16_scale[((size_t)cur.clip * p.nsub + cur.band) * 2] : 1.0f;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // The 14 k-steps run half by half (tap rows 0..3, then 4..6), tap column by tap column, and A FRAGMENTS ARE SHARED BETWEEN
      // TAP COLUMNS: M-tile t at tap column c is the patch units of columns 10 g + 2 t + dc + c -- the very registers M-tile t + 1
      // held at column c - 2.  So per half and column parity there are 8 (7) distinct fragments, "virtual M-tiles" j = t + (c >> 1),
      // each read from LDS ONCE: 15 reads per half instead of 35; with the weights, 116 ds_read_b128 per tile instead of 196 (the
      // LDS data path was busy 55-70 % of the kernel).  Fragments of the NEXT step are requested while the MFMAs of this one run;
      // every address is a lane base (half) plus a compile-time offset.
      h16x8 Rh[2][2][8], Rl[2][2][TERMS == 3 ? 8 : 1], bh[2][NT], bl[2][NT];
      auto read_for = [&](int q) __attribute__((always_inline)) {   // what sequence step q needs and no register holds yet
        const int half = q / 7, c = q % 7, par = c & 1, st = 2 * c + half, buf = q & 1;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          bh[buf][n] = wres[((st * NT + n) * HL + 0) * 64 + lane];
          if (TERMS == 3) bl[buf][n] = wres[((st * NT + n) * HL + 1) * 64 + lane];
        }
        const h16x8* pa = phi + (half ? base1 : base0) + par;
        const h16x8* pl = plo + (half ? base1 : base0) + par;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (c < 2 ? j < MT : j == MT - 1 + (c >> 1)) {
            Rh[half][par][j] = pa[2 * j];
            if (TERMS == 3) Rl[half][par][j] = pl[2 * j];
          }
      };
      read_for(0);
      static_for<kF16StepsE>([&](auto Q_) __attribute__((always_inline)) {
        constexpr int q = decltype(Q_)::value;
        constexpr int half = q / 7, c = q % 7, par = c & 1, sh = c >> 1, cu = q & 1;
        if constexpr (q + 1 < kF16StepsE) read_for(q + 1);
        prefetch_piece(q);                                        // 6 (+ 6) 16-byte loads of the next tile, in the first six steps
        // pin the reads IN FRONT of this step's MFMAs: left to itself the scheduler sinks them into the middle of the step and the
        // next step opens with an exposed LDS round trip (counted lgkmcnt waits two MFMAs after the reads)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if (TERMS == 3) {
              acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Rl[half][par][TERMS == 3 ? t + sh : 0], bh[cu][n], acc[t][n], 0, 0, 0);  // small terms first
              acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Rh[half][par][t + sh], bl[cu][n], acc[t][n], 0, 0, 0);
            }
            acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Rh[half][par][t + sh], bh[cu][n], acc[t][n], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);   // keep the next step's reads ahead of this step's MFMAs, per step
      });
    }
    stage();   // the next tile's patch (this tile's LDS reads are all consumed)
    // (the constants are first TOUCHED here: without this the scheduler folds the weight pre-scale into the affine right after
    // the loads and the wave sits out their round trip at the head of the tile)
#pragma unroll
    for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(e_ac[n].x), "+v"(e_ac[n].y), "+v"(e_winv[n]));
    asm volatile("" : "+v"(f16s));
    {   // epilogue: y = A * acc + C, ReLU, max over the window -- accumulator slot (t, r) is position (row r >> 1, column
        // 10 g + 2 t + (r & 1)): columns 0..4 of the lane's ten are pooling window 0, columns 5..9 window 1
      const int j = lane & 15, g = lane >> 4;
      float* orow = p.out + ((size_t)cur.clip * p.nsub + cur.band) * C::COUT * p.out_rows * p.out_cols +
                    (size_t)cur.tr * p.out_cols;
      const int nh = p.pool_h == 1 ? 2 : 1;   // pool height 1 (16-mel sub-bands): two 1 x 5 windows (tile rows 0 and 1) per window column block
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int ch = n * 16 + j;
        float2 ac = e_ac[n];
        ac.x *= e_winv[n];  // undo the weight pre-scale
#pragma unroll
        for (int wv = 0; wv < 2; ++wv) {
          const int pc = 8 * cur.tc + 2 * g + wv;
#pragma unroll
          for (int hr = 0; hr < 2; ++hr) {
            if (hr < nh) {
              float m = 0.f;   // ReLU floor
#pragma unroll
              for (int cc = 5 * wv; cc < 5 * wv + 5; ++cc)
#pragma unroll
                for (int dr = 0; dr < 2; ++dr) {
                  const bool mine = nh == 1 || dr == hr;
                  const float v = fmaf(acc[cc >> 1][n][2 * dr + (cc & 1)], ac.x, ac.y);
                  m = mine ? fmaxf(m, v) : m;
                }
              const int orow_i = nh == 1 ? cur.tr : 2 * cur.tr + hr;   // output row
              if (pc < p.out_cols && orow_i < p.out_rows) {
                if (p.out) orow[(size_t)ch * p.out_rows * p.out_cols + (size_t)(orow_i - cur.tr) * p.out_cols + pc] = m;
                if (out_hi) {  // conv2's split-precision input: [clip][band][row][col][32 ch] f16, hi and lo, range-scaled
                  const size_t o = ((((size_t)cur.clip * p.nsub + cur.band) * p.out_rows + orow_i) * p.out_cols + pc) * 32 + ch;
                  const float ms = m * f16s;
                  const _Float16 h = (_Float16)ms;
                  out_hi[o] = h;
                  if (TERMS == 3) out_lo[o] = (_Float16)(ms - (float)h);
                }
              }
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// OPT-IN conv2 on the f16 matrix cores, 3-term split precision (mst_encoder_set_precision(enc, 2)).
// Input: conv1's pooled activations as f16 hi/lo, channel-minor.  A k-step is 4 taps x 8 input channels; the 32
// input channels are processed in 4 chunks of 8 (13 k-steps each; the chunk's 104 KB of hi/lo weight fragments are
// shared by the 8 waves, every wave has a private 14x14-position patch).  The input arrives range-scaled per
// (clip, band) by f16_scale_kernel's power of two, undone in the epilogue: no activation magnitude saturates f16.
// ------------------------------------------------------------------------------------------
// MODE 1 (training forward, TERMS = 1): raw output + bias in the 8 x 8-tile accumulator order and the batch-statistics
// sums over the valid positions, as conv_kernel<2, 2, 1> leaves them (tiles past the last row are computed, not stored
// in the sums: f16 MFMAs are cheap enough that the 2-row strip kernel of the fp32 path is not needed).
template <int TERMS, int MODE = 0>
__global__ __launch_bounds__(kConvThreads) void conv2_f16x3_kernel(const ConvParams p, const h16x8* __restrict__ in_hi,
                                                                  const h16x8* __restrict__ in_lo,
                                                                  const h16x8* __restrict__ wfrag16) {
  constexpr int MT = 4, NT = 4, NCH = 4, PR = 14, PC = 14, NPOS = PR * PC;
  constexpr int HLW = TERMS == 3 ? 2 : 1;             // the fragment buffer holds hi and lo; TERMS = 1 stages the hi vectors only
  constexpr int WVG = kF16Steps * NT * 2 * 64;        // h16x8 vectors per weight chunk in global memory (hi/lo interleaved per (step, nt))
  constexpr int WV = kF16Steps * NT * HLW * 64;       // ... staged into LDS (6656 = 13 per thread, or 3328)
  constexpr int NWF = (WV + kConvThreads - 1) / kConvThreads;
  constexpr int NPF = (NPOS + 63) / 64;               // position vectors per lane (hi and lo each)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  h16x8* wres = reinterpret_cast<h16x8*>(smem);       // [WV]
  h16x8* phi = wres + WV + wave * 2 * NPOS;
  h16x8* plo = phi + NPOS;

  const int G = gridDim.x;
  const int wg = mst::xcd_remap(blockIdx.x, G);
  const int total_sets = p.nsub * p.sets_per_band;
  // eval: sets dealt round-robin over the workgroups.  Training (MODE >= 1): every workgroup takes a CONTIGUOUS range of sets, so
  // that its band -- and with it the flush of the batch-statistics sums, 256 integer atomics per wave -- changes once or twice
  // per launch instead of every third tile (the round-robin deal spent more time in those atomics than in the MFMAs)
  const int s_first = MODE >= 1 ? (int)((long long)wg * total_sets / G) : wg;
  const int my_sets = MODE >= 1 ? (int)((long long)(wg + 1) * total_sets / G) - s_first
                                : (wg < total_sets ? (total_sets - wg + G - 1) / G : 0);
  const int nq = my_sets * NCH;
  const int tpb = p.tiles_r * p.tiles_c;
  auto decode = [&](int q) __attribute__((always_inline)) {
    const int s = MODE >= 1 ? min(s_first + q / NCH, total_sets - 1) : wg + (q / NCH) * G;
    Tile t;
    t.band = min(s / p.sets_per_band, p.nsub - 1);
    const int idx = (s - t.band * p.sets_per_band) * kConvWaves + wave;
    t.valid = (q < nq) && idx < p.B * tpb;
    t.clip = t.valid ? idx / tpb : 0;
    const int ti = t.valid ? idx - t.clip * tpb : 0;
    t.tc = ti / p.tiles_r;
    t.tr = ti - t.tc * p.tiles_r;
    return t;
  };
  const int kq = lane >> 4, ai = lane & 15, ag = ai >> 2, areg = ai & 3;
  int abase[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) abase[t] = (4 * (ag >> 1) + t) * PC + 4 * (ag & 1) + areg;
  int toff[kF16Steps];
#pragma unroll
  for (int st = 0; st < kF16Steps; ++st) {
    const int tap = min(4 * st + kq, 48);
    toff[st] = (tap / 7) * PC + tap % 7;
  }

  // split-precision TRAINING forward (TERMS 3, MODE >= 1): the 52 registers that would hold the next chunk's weights through the
  // MFMA phase do not exist next to the hi / lo fragments and the statistics (53 spilled registers, 2.4 ms instead of ~1.6):
  // this instantiation fetches the chunk's weights AFTER the phase, straight through to LDS (their latency is exposed once per
  // chunk, the registers are the fragments' own)
  constexpr bool LATEW = TERMS == 3 && MODE >= 1;
  h16x8 wreg[LATEW ? 1 : NWF], ph[NPF], pl[NPF];
  auto prefetch = [&](int q, const Tile& t) __attribute__((always_inline)) {
    const int chunk = q % NCH;
    const h16x8* wsrc = wfrag16 + ((size_t)t.band * NCH + chunk) * WVG;
    if constexpr (!LATEW) {
#pragma unroll
      for (int i = 0; i < NWF; ++i) {
        if constexpr (HLW == 2) {   // WV is a multiple of the block size: one base address + immediate offsets
          wreg[i] = wsrc[tid + kConvThreads * i];
        } else {
          const int k = min(tid + kConvThreads * i, WV - 1);
          wreg[i] = wsrc[((k >> 6) * 2) * 64 + (k & 63)];
        }
      }
    }
    const int row0 = 8 * t.tr - 3, col0 = 8 * t.tc - 3;
    const size_t plane = ((size_t)t.clip * p.nsub + t.band) * p.in_rows;
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int e = lane + 64 * i;
      const int r = e / PC, c = e - r * PC;
      const int rin = row0 + r, cin = col0 + c;
      const int rc = min(max(rin, 0), p.in_rows - 1), cl = min(max(cin, 0), p.in_cols - 1);
      const bool ok = t.valid && e < NPOS && rin == rc && cin == cl;
      const size_t v = ((plane + rc) * p.in_cols + cl) * 4 + chunk;
      const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      const h16x8 a = in_hi[v];
      ph[i] = ok ? a : z;
      if (TERMS == 3) {
        const h16x8 b = in_lo[v];
        pl[i] = ok ? b : z;
      }
    }
  };

  f32x4 acc[MT][NT];
  double st[NT][2];
  int st_band = -1;
#pragma unroll
  for (int n = 0; n < NT; ++n) st[n][0] = st[n][1] = 0.0;
  Tile cur{}, nxt = decode(0);
  if (nq > 0) prefetch(0, nxt);
  for (int q = 0; q < nq; ++q) {
    __syncthreads();  // every wave is done with the previous chunk's weights
    if constexpr (LATEW) {
      const h16x8* wsrc = wfrag16 + ((size_t)nxt.band * NCH + q % NCH) * WVG;   // (nxt = the tile of this q)
#pragma unroll
      for (int i = 0; i < NWF; ++i) wres[tid + kConvThreads * i] = wsrc[tid + kConvThreads * i];
    } else {
#pragma unroll
      for (int i = 0; i < NWF; ++i)
        if (WV % kConvThreads == 0 || tid + kConvThreads * i < WV) wres[tid + kConvThreads * i] = wreg[i];
    }
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int e = lane + 64 * i;
      if (e < NPOS) {
        phi[e] = ph[i];
        if (TERMS == 3) plo[e] = pl[i];
      }
    }
    __syncthreads();
    cur = nxt;
    nxt = decode(q + 1);
    prefetch(q + 1, nxt);   // lands while this chunk is on the MFMAs; past the end it re-reads valid memory
    const int chunk = q % NCH;
    if (chunk == 0) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // training forward, last tile row of a plane whose height is 8 k + 1 or 8 k + 2 (rows 8, 9 of 10): only M-tiles 0, 1 hold
    // rows that exist.  Two straight-line instantiations of the k-step loop, chosen per tile (wave-uniform): a branch around
    // single MFMAs inside one loop breaks the software pipeline.  (A macro, not a lambda: capturing the accumulator arrays
    // by reference cost registers -- 34 spilled VGPRs in the eval instantiation.)
#define MST_CONV2_F16_KSTEPS(MTX)                                                                                         \
  _Pragma("unroll") for (int st = 0; st < kF16Steps; ++st) {                                                              \
    h16x8 ah[MTX], al[MTX], bh[NT], bl[NT];                                                                               \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                                      \
      bh[n] = wres[((st * NT + n) * HLW + 0) * 64 + lane];                                                                \
      if (TERMS == 3) bl[n] = wres[((st * NT + n) * HLW + 1) * 64 + lane];                                                \
    }                                                                                                                     \
    _Pragma("unroll") for (int t = 0; t < MTX; ++t) {                                                                     \
      ah[t] = phi[abase[t] + toff[st]];                                                                                   \
      if (TERMS == 3) al[t] = plo[abase[t] + toff[st]];                                                                   \
    }                                                                                                                     \
    _Pragma("unroll") for (int t = 0; t < MTX; ++t)                                                                       \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                                    \
        if (TERMS == 3) {                                                                                                 \
          acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh[n], acc[t][n], 0, 0, 0);                          \
          acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl[n], acc[t][n], 0, 0, 0);                          \
        }                                                                                                                 \
        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh[n], acc[t][n], 0, 0, 0);                            \
      }                                                                                                                   \
  }
    const bool low_rows = MODE >= 1 && __builtin_amdgcn_readfirstlane(cur.tr) >= 1 &&
                          8 * __builtin_amdgcn_readfirstlane(cur.tr) + 2 >= p.raw_rows;
    if constexpr (TERMS == 1 && MODE == 0) {   // the fragments of step st + 1 are read from LDS while the MFMAs of step st run
      // (double-buffered registers; with the low parts of TERMS = 3, or the statistics of the training modes, the second set
      // does not fit the register file)
      h16x8 a[2][MT], b[2][NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[0][n] = wres[n * 64 + lane];
#pragma unroll
      for (int t = 0; t < MT; ++t) a[0][t] = phi[abase[t] + toff[0]];
#pragma unroll
      for (int st = 0; st < kF16Steps; ++st) {
        const int cu = st & 1, nx = cu ^ 1;
        if (st + 1 < kF16Steps) {
#pragma unroll
          for (int n = 0; n < NT; ++n) b[nx][n] = wres[((st + 1) * NT + n) * 64 + lane];
#pragma unroll
          for (int t = 0; t < MT; ++t) a[nx][t] = phi[abase[t] + toff[st + 1]];
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cu][t], b[cu][n], acc[t][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep the next step's reads ahead of this step's MFMAs, per step
      }
    } else if (MODE >= 1 && low_rows) {
      MST_CONV2_F16_KSTEPS(2)
    } else {
      MST_CONV2_F16_KSTEPS(MT)
    }
#undef MST_CONV2_F16_KSTEPS
    if (MODE >= 1 && chunk == NCH - 1 && cur.valid) {
      const int j = lane & 15, g = lane >> 4;
      if (cur.band != st_band) {
        if (st_band >= 0) flush_stats<NT, 64>(st, p.stats, st_band, lane);
        st_band = cur.band;
      }
      const float inv_s = p.f16_scale ? p.f16_scale[((size_t)cur.clip * p.nsub + cur.band) * 2 + 1] : 1.0f;
      const size_t unit0 = ((((size_t)cur.clip * p.nsub + cur.band) * p.acc_tr + cur.tr) * p.acc_tc + cur.tc) * (size_t)(NT * 64);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float b = p.bias[cur.band * 64 + n * 16 + j];
        const float wi = inv_s * p.f16_winv[cur.band * 64 + n * 16 + j];   // input range scale and weight pre-scale, exact powers of two
        const size_t unit = unit0 + (size_t)(n * 64 + lane);
        const float sy = MODE == 2 ? p.y_scale[cur.band * 2] : 1.f, isy = MODE == 2 ? p.y_scale[cur.band * 2 + 1] : 1.f;
        // the tile's 16 values are summed in fp32 first and folded into the double accumulators once per tile: on this
        // MFMA-light kernel 128 fp64 operations per lane and tile cost more than the convolution (3.07 -> 1.98 ms without them)
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          f32x4 v = acc[t][n];
          h16x4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = fmaf(v[r], wi, b);
            if constexpr (MODE == 2) {
              h[r] = to_f16_sat(v[r] * sy);
              v[r] = (float)h[r] * isy;
            }
            const int row = 8 * cur.tr + 4 * (g >> 1) + t, col = 8 * cur.tc + 4 * (g & 1) + r;
            if (row < p.raw_rows && col < p.raw_cols) ps += v[r], pq = fmaf(v[r], v[r], pq);
          }
          if constexpr (MODE == 2) reinterpret_cast<h16x4*>(p.yraw16)[yvec(unit, t, MT)] = h;   // t-major: whole-line stores
          else reinterpret_cast<f32x4*>(p.yraw)[yvec(unit, t, MT)] = v;
        }
        st[n][0] += (double)ps, st[n][1] += (double)pq;
      }
    }
    if (MODE == 0 && chunk == NCH - 1 && cur.valid) {
      const int j = lane & 15, g = lane >> 4;
      const float2* aff = p.aff + ((size_t)cur.clip * p.nsub + cur.band) * 64;
      const float inv_s = p.f16_scale ? p.f16_scale[((size_t)cur.clip * p.nsub + cur.band) * 2 + 1] : 1.0f;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int ch = n * 16 + j;
        float2 ac = aff[ch];
        ac.x *= inv_s * p.f16_winv[cur.band * 64 + ch];   // undo the input's range scale and the weight pre-scale (exact powers of two)
        float m = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, fmaf(acc[t][n][r], ac.x, ac.y));
        const int pr = 2 * cur.tr + (g >> 1), pc = 2 * cur.tc + (g & 1);
        if (pr < p.out_rows && pc < p.out_cols)
          p.out[(((size_t)cur.clip * p.nsub + cur.band) * 64 + ch) * p.out_rows * p.out_cols + (size_t)pr * p.out_cols + pc] = m;
      }
    }
  }
  if (MODE >= 1 && st_band >= 0) flush_stats<NT, 64>(st, p.stats, st_band, lane);
}

// ------------------------------------------------------------------------------------------
// Attention scores: s[b][t] = w2 . tanh(W1 x[b,:,t] + b1) + b2      (model.py:198-200)
// fp32-MFMA GEMM.  One workgroup = 16 frames; its 4 waves split the 256 hidden units (4 N-tiles each) and
// combine their partial dot products through LDS.  W1 is pre-swizzled into B-fragment order.
// ------------------------------------------------------------------------------------------
struct AttnParams {
  const float* x;       // pool_in [B][C][W]
  const float* w1frag;  // [C/4][16][64]
  const float *b1, *w2;
  const float* b2;      // device [1]
  float* scores;        // [B][W]
  int B, C, W, A;
};

#ifndef MST_ATTN_MT
#define MST_ATTN_MT 1
#endif
#ifndef MST_ATTN_U
#define MST_ATTN_U 2
#endif
constexpr int kAttnMT = MST_ATTN_MT;   // M-tiles (16 frames each) per workgroup: every weight fragment fetched from L2 feeds kAttnMT MFMAs
__global__ __launch_bounds__(256) void attn_scores_kernel(const AttnParams p) {
  __shared__ float part_s[4][16 * kAttnMT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mt = blockIdx.x;
  const int M = p.B * p.W;
  const int kq = lane >> 4, i = lane & 15;
  const float* xa[kAttnMT];
  bool ok[kAttnMT];
#pragma unroll
  for (int q = 0; q < kAttnMT; ++q) {
    const int m = (mt * kAttnMT + q) * 16 + i;
    ok[q] = m < M;
    const int b = ok[q] ? m / p.W : 0, t = ok[q] ? m % p.W : 0;
    xa[q] = p.x + ((size_t)b * p.C + kq) * p.W + t;
  }
  f32x4 acc[kAttnMT][4];
#pragma unroll
  for (int q = 0; q < kAttnMT; ++q)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[q][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int steps = p.C / 4;
  const float* wf = p.w1frag + (size_t)(wave * 4) * 64 + lane;
  // software-pipelined over batches of kAttnU k-steps: the loads of batch j + 1 are in flight while the MFMAs of batch j run
  // (the loop is load-latency-bound: one wave per SIMD and a half, 352 dependent-free k-steps)
  constexpr int kAttnU = MST_ATTN_U;
  float a[2][kAttnU][kAttnMT], bb[2][kAttnU][4];
  auto load_batch = [&](int sb, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < kAttnU; ++u) {
      const int s = min(sb + u, steps - 1);
#pragma unroll
      for (int q = 0; q < kAttnMT; ++q)
        a[buf][u][q] = (ok[q] && sb + u < steps) ? xa[q][(size_t)s * 4 * p.W] : 0.f;   // a zero A fragment makes a padded step a no-op
#pragma unroll
      for (int n = 0; n < 4; ++n) bb[buf][u][n] = wf[((size_t)s * 16 + n) * 64];
    }
  };
  auto mfma_batch = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < kAttnU; ++u)
#pragma unroll
      for (int q = 0; q < kAttnMT; ++q)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][u][q], bb[buf][u][n], acc[q][n], 0, 0, 0);
  };
  load_batch(0, 0);
  for (int sb = 0; sb < steps; sb += 2 * kAttnU) {   // (a batch past the end loads clamped addresses and multiplies zeros)
    load_batch(sb + kAttnU, 1);
    mfma_batch(0);
    load_batch(sb + 2 * kAttnU, 0);
    mfma_batch(1);
  }
  // rows 4*kq + r of an M-tile live in lane group kq; columns (wave*4 + n)*16 + i
#pragma unroll
  for (int q = 0; q < kAttnMT; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float part = 0.f;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int h = (wave * 4 + n) * 16 + i;
        part = fmaf(p.w2[h], tanhf(acc[q][n][r] + p.b1[h]), part);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      if (i == 0) part_s[wave][16 * q + 4 * kq + r] = part;
    }
  __syncthreads();
  if (threadIdx.x < 16 * kAttnMT) {
    const int mr = mt * 16 * kAttnMT + threadIdx.x;
    if (mr < M)
      p.scores[mr] = ((part_s[0][threadIdx.x] + part_s[1][threadIdx.x]) + (part_s[2][threadIdx.x] + part_s[3][threadIdx.x])) + p.b2[0];
  }
}

// softmax over frames + weighted sum: pooled[b][c] = sum_t softmax(s[b])[t] * x[b][c][t]   (model.py:201-206)
struct PoolParams {
  const float* x;       // [B][C][W]
  const float* scores;  // [B][W]
  float* pooled;        // [B][C]
  int C, W;
};

__global__ __launch_bounds__(256) void attn_pool_kernel(const PoolParams p) {
  extern __shared__ float sm[];
  float* w = sm;  // [W] normalised weights
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int t = tid; t < p.W; t += 256) mx = fmaxf(mx, p.scores[(size_t)b * p.W + t]);
  mx = mst::wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int t = tid; t < p.W; t += 256) {
    const float e = expf(p.scores[(size_t)b * p.W + t] - mx);
    w[t] = e;
    sum += e;
  }
  sum = mst::wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
  // this block's slice of channels: 16 lanes per channel, 4 channels per wave pass
  const int c_per_blk = (p.C + gridDim.y - 1) / gridDim.y;
  const int c0 = blockIdx.y * c_per_blk, c1 = min(p.C, c0 + c_per_blk);
  const float* xb = p.x + (size_t)b * p.C * p.W;
  const int sub = lane >> 4, l16 = lane & 15;
  for (int c = c0 + wave * 4 + sub; c < c1 + 3; c += 16) {
    float v = 0.f;
    if (c < c1)
      for (int t = l16; t < p.W; t += 16) v = fmaf(xb[(size_t)c * p.W + t], w[t] * inv, v);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (l16 == 0 && c < c1) p.pooled[(size_t)b * p.C + c] = v;
  }
}

// projection + ReLU: emb[b][e] = relu(bp[e] + sum_c Wp[e][c] pooled[b][c])   (model.py:208-209)
// fp32-MFMA GEMM, one wave = one 16-column tile of E x up to 8 row tiles of clips; Wp in B-fragment order.
struct ProjParams {
  const float* pooled;  // [B][C]
  const float* wfrag;   // [C/4][E/16][64]
  const float* bias;    // [E]
  float* emb;           // [B][E]
  int B, C, E;
};

template <int MT>
__global__ __launch_bounds__(512) void proj_kernel(const ProjParams p) {
  // one workgroup = one 16-column tile of E x MT row tiles of clips; its 8 waves split K (split-K inside the
  // workgroup, LDS reduce); every wave keeps 4 k-steps of loads in flight (the plain loop was latency-bound)
  extern __shared__ float red[];  // [8 waves][MT * 4 regs][64 lanes]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nt = blockIdx.x, NTE = p.E / 16;
  const int kq = lane >> 4, i = lane & 15;
  const int m0 = blockIdx.y * (16 * MT);
  f32x4 acc[MT];
  const float* ap[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    ap[t] = p.pooled + (size_t)min(m0 + t * 16 + i, p.B - 1) * p.C + kq;   // rows past B are computed and dropped
  }
  const float* wf = p.wfrag + (size_t)nt * 64 + lane;
  const int steps = p.C / 4;
  const int s0 = wave * steps / 8, s1 = (wave + 1) * steps / 8;
  for (int sb = s0; sb < s1; sb += 4) {
    float bb[4], a[4][MT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = min(sb + u, s1 - 1);
      bb[u] = (sb + u < s1) ? wf[(size_t)s * NTE * 64] : 0.f;   // a zero B fragment makes the padded step a no-op
#pragma unroll
      for (int t = 0; t < MT; ++t) a[u][t] = ap[t][4 * s];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], bb[u], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * (4 * MT) + t * 4 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  const int e = nt * 16 + i;
  const float be = p.bias[e];
  for (int t = wave; t < MT; t += 8) {  // wave w finalises tiles w, w+8, ...
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[(w * (4 * MT) + t * 4 + r) * 64 + lane];
      const int m = m0 + t * 16 + 4 * kq + r;
      if (m < p.B) p.emb[(size_t)m * p.E + e] = fmaxf(v + be, 0.f);
    }
  }
}

template <int MT>
void launch_proj(const ProjParams& pj, int ygrid, hipStream_t st) {
  hipLaunchKernelGGL((proj_kernel<MT>), dim3(pj.E / 16, ygrid), dim3(512), (size_t)8 * 4 * MT * 64 * sizeof(float), st, pj);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// conv1 + BN(eval) + FiLM + ReLU + MaxPool(SUB, 5) for the first-pool heights the MFMA kernels have no tile for
// (split_size >= 30: SUB = split_size // 10 >= 3, src/model.py:111-117).  Coverage of the reference's geometry, not the fast
// path: fp32 FMAs on the vector unit, one workgroup = (clip, band, 8 pooled columns) holding the band's 392 x 32 weights and
// the whole (split + 6) x 46 x 8 input patch in LDS; thread = (output channel, pooled column), walking the pooled rows.
// Reads the weights from the MFMA fragment table (conv_fragments: [cin / 4][tap][n / 16][lane = 16 (cin % 4) + n % 16]).
// ------------------------------------------------------------------------------------------
constexpr int kGenCols = 8, kGenPC = 5 * kGenCols + 6;
__global__ __launch_bounds__(256) void conv1_generic_kernel(const ConvParams p, int sub) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* wl = gsm;                    // [392 (cin, tap)][32 cout]
  float* patch = gsm + 392 * 32;      // [8][split + 6][kGenPC]
  const int tid = threadIdx.x, co = tid & 31, pwl = tid >> 5;
  const int clip = blockIdx.y / p.nsub, band = blockIdx.y - clip * p.nsub;
  const int pw0 = blockIdx.x * kGenCols, PR = p.in_rows + 6;
  constexpr int WBP = ConvGeom<1, 2>::WBP;
  for (int i = tid; i < 392 * 32; i += 256) {
    const int k = i >> 5, c = i & 31, ci = k / 49, tap = k - ci * 49;
    wl[i] = p.wfrag[((size_t)band * 2 + (ci >> 2)) * WBP + ((size_t)tap * 2 + (c >> 4)) * 64 + 16 * (ci & 3) + (c & 15)];
  }
  const float* src = p.in + (size_t)clip * p.in_clipstride + (size_t)band * p.in_bandoff;
  for (int i = tid; i < 8 * PR * kGenPC; i += 256) {
    const int ci = i / (PR * kGenPC), r = (i / kGenPC) % PR, c = i % kGenPC;
    const int rin = r - 3, cin = 5 * pw0 + c - 3;
    patch[i] = (rin >= 0 && rin < p.in_rows && cin >= 0 && cin < p.in_cols) ? src[(size_t)ci * p.in_cstride + (size_t)rin * p.in_cols + cin] : 0.f;
  }
  __syncthreads();
  const int pw = pw0 + pwl;
  if (pw >= p.out_cols) return;
  const float2 ac = p.aff[((size_t)clip * p.nsub + band) * 32 + co];
  float* dst = p.out + (((size_t)clip * p.nsub + band) * 32 + co) * p.out_rows * p.out_cols + pw;
  for (int ph = 0; ph < p.out_rows; ++ph) {
    float best = 0.f;   // ReLU: max(0, .) folded into the running maximum
    for (int r = 0; r < sub; ++r) {
      float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      for (int ci = 0; ci < 8; ++ci)
        for (int ky = 0; ky < 7; ++ky) {
          const float* row = patch + (ci * PR + ph * sub + r + ky) * kGenPC + 5 * pwl;
          float x[11];
#pragma unroll
          for (int j = 0; j < 11; ++j) x[j] = row[j];
          const float* wk = wl + ((ci * 49 + ky * 7) << 5) + co;
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) {
            const float w = wk[kx << 5];
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[c] = fmaf(w, x[c + kx], acc[c]);
          }
        }
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const float y = fmaf(ac.x, acc[c], ac.y);
        best = (y != y) ? y : fmaxf(best, y);   // a NaN stays visible
      }
    }
    dst[(size_t)ph * p.out_cols] = best;
  }
}

struct mst_encoder {
  mst_encoder_config cfg{};
  int sub = 0, H1 = 0, FD = 0, C = 0;
  float *w1frag = nullptr, *w2frag = nullptr;
  float *s1 = nullptr, *t1 = nullptr, *s2 = nullptr, *t2 = nullptr;
  float *w0t = nullptr, *b0 = nullptr, *w3t = nullptr, *b3 = nullptr, *hwt = nullptr, *hb = nullptr;
  float *att0frag = nullptr, *att0_b = nullptr, *att2_w = nullptr, *projfrag = nullptr, *proj_b = nullptr;
  float* att2_b = nullptr;    // device [1]
  int num_cus = 256;
  float *f16_wsc1e = nullptr, *f16_wsc2e = nullptr, *w2norm_e = nullptr;   // scratch of the device-side refresh (mst_encoder_update_params)
  void* w1frag16 = nullptr;   // conv1 weights as f16 hi/lo MFMA B fragments (opt-in split-precision path)
  void* w1frag16e = nullptr;  // the same weights in conv1_f16e_kernel's k order (14 steps: one tap column x four tap rows)
  void* w2frag16 = nullptr;   // conv2 likewise: [band][4 chunks][13 steps][4 nt][hi/lo][lane][8]
  float* w1norm = nullptr;    // [nsub][32] L1 norm of every conv1 filter (range bound of the f16 path)
  float *f16_winv1 = nullptr, *f16_winv2 = nullptr;   // [nsub][32], [nsub][64]: inverse pre-scale of the f16 weight fragments
  int conv1_f16x3 = 0;        // 0 exact fp32, 1 conv1 f16x3, 2 conv1 + conv2 f16x3, 3 conv1 + conv2 plain f16 (amp)
  // un-folded parameters for the training forward (batch-statistics BatchNorm)
  float *c1b = nullptr, *bn1w = nullptr, *bn1b = nullptr, *c2b = nullptr, *bn2w = nullptr, *bn2b = nullptr;
  float* w2dfrag = nullptr;   // conv2 input-gradient weight fragments [nsub][16][WBP] (refreshed by update_trunk_params)
  // f16-operand training (mst_encoder_set_train_precision): fragments and pre-scales rebuilt on the device every step
  int train_f16 = 0;
  void* w2dfrag16 = nullptr;  // conv2 input-gradient fragments, hi only: [nsub][8 chunks][13 steps][2 nt][lane][8]
  float *f16_wsc1 = nullptr, *f16_wsc2 = nullptr, *f16_wsc2d = nullptr, *f16_winv2d = nullptr;   // pre-scales 2^k and the dgrad inverse
  float* w2norm = nullptr;    // [nsub][64] L1 norms of the conv2 filters (range bound of the stored f16 conv2 outputs, mode 1)
};

namespace {

// training precision modes (mst_encoder_set_train_precision): 0 fp32 MFMA; 1 f16 operands; 2 three-term split-precision f16
// (fp32-equivalent) -- both on the f16 matrix cores, forward and backward (encoder_f16train.inc)
inline bool train_fwd16(const mst_encoder* e) { return e->train_f16 != 0; }
inline bool train_bwd16(const mst_encoder* e) { return e->train_f16 != 0; }

struct WsLayout {
  size_t film, aff1, aff2, pool1, pool1_h16, pool1_l16, f16scale, xmax, pool_in, scores, pooled, total;
  int W1, W2;
};

WsLayout ws_layout(const mst_encoder* e, int B, int frames) {
  WsLayout L{};
  const int ns = e->cfg.n_subbands;
  L.W1 = frames / 5;
  L.W2 = L.W1 / 4;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += mst::align_up(bytes, 256);
    return at;
  };
  L.film = take((size_t)B * ns * 192 * 4);
  L.aff1 = take((size_t)B * ns * 32 * 8);
  L.aff2 = take((size_t)B * ns * 64 * 8);
  L.pool1 = take((size_t)B * ns * 32 * e->H1 * L.W1 * 4);
  L.pool1_h16 = take(e->conv1_f16x3 >= 2 ? (size_t)B * ns * 32 * e->H1 * L.W1 * 2 : 0);
  L.pool1_l16 = take(e->conv1_f16x3 == 2 ? (size_t)B * ns * 32 * e->H1 * L.W1 * 2 : 0);
  L.f16scale = take(e->conv1_f16x3 >= 2 ? (size_t)B * ns * 2 * 4 : 0);
  L.xmax = take(e->conv1_f16x3 >= 2 ? (size_t)B * 4 : 0);
  L.pool_in = take((size_t)B * e->C * L.W2 * 4);
  L.scores = take((size_t)B * L.W2 * 4);
  L.pooled = take((size_t)B * e->C * 4);
  L.total = o;
  return L;
}

std::vector<float> transpose(const float* w, int rows, int cols) {  // [rows][cols] -> [cols][rows]
  std::vector<float> t((size_t)rows * cols);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = w[(size_t)r * cols + c];
  return t;
}

// conv weights [nsub][COUT][CIN][7][7] -> [nsub][CIN/4][WCHP] with chunk layout [tap][nt][lane]
std::vector<float> conv_fragments(const float* w, int nsub, int cout, int cin, int wchp) {
  const int nt = cout / 16, nch = cin / 4;
  std::vector<float> f((size_t)nsub * nch * wchp, 0.f);
  for (int b = 0; b < nsub; ++b)
    for (int ch = 0; ch < nch; ++ch)
      for (int tap = 0; tap < 49; ++tap)
        for (int n = 0; n < nt; ++n)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = n * 16 + (lane & 15), ci = 4 * ch + (lane >> 4);
            f[((size_t)b * nch + ch) * wchp + ((size_t)tap * nt + n) * 64 + lane] =
                w[(((size_t)b * cout + co) * cin + ci) * 49 + tap];
          }
  return f;
}

// ------------------------------------------------------------------------------------------
// Training forward (first half): train-mode BatchNorm needs the statistics of the whole batch before
// anything downstream of the convolution can be computed, so the fused eval kernels are split in two:
//   conv (MODE 1)  ->  yraw (accumulator order) + per-(band, channel) sums        [conv kernels above]
//   bn_fold_kernel ->  batch mean / 1/std, and the per-(clip, band, channel) affine  A*y + C  that carries
//                      BatchNorm(batch statistics) and FiLM, as in eval
//   apply_kernel   ->  affine + ReLU + max-pool per lane (the pooling window is the lane's own 10 / 16 values)
// yraw is kept: the backward pass recomputes everything between y and the pooled output from it.
// ------------------------------------------------------------------------------------------
struct FoldParams {
  const mst::DetAcc* stats;   // [nsub][COUT][2]
  const float *bn_w, *bn_b;  // [nsub][COUT]
  const float* film;       // [B][nsub*192]
  float2* aff;             // [B][nsub][COUT]
  float2* bnstat;          // [nsub][COUT] (batch mean, 1/sqrt(biased var + eps))
  double count;            // positions per (band, channel): B * rows * cols
  float eps;
  int nsub, cout, goff, boff;   // FiLM gamma / beta offsets inside a band's 192 values
  // cross-rank statistics (phased calls): the clip count comes from the device word that was summed over the ranks together
  // with the statistics (the word after stats[nsub][COUT][2]), times per_clip = rows * cols -- ranks may hold DIFFERENT
  // numbers of clips (ragged last batch) and still agree on mean / variance bit for bit.  NULL: `count` from the host.
  const long long* clips;
  double per_clip;
};

__global__ void bn_fold_kernel(const FoldParams p) {   // grid (nsub, B), block COUT
  const int band = blockIdx.x, b = blockIdx.y, ch = threadIdx.x;
  const size_t i = (size_t)band * p.cout + ch;
  const double count = p.clips ? (double)p.clips[0] * p.per_clip : p.count;
  const double mean = mst::det_get(p.stats[i * 2]) / count;
  const double var0 = mst::det_get(p.stats[i * 2 + 1]) / count - mean * mean;
  const double var = var0 != var0 ? var0 : fmax(var0, 0.0);   // (fmax would turn a poisoned, NaN statistic into 0)
  const double invstd = 1.0 / sqrt(var + (double)p.eps);
  if (b == 0) p.bnstat[i] = make_float2((float)mean, (float)invstd);
  const float* fl = p.film + ((size_t)b * p.nsub + band) * 192;
  const double g = fl[p.goff + ch], be = fl[p.boff + ch];
  const double sc = (double)p.bn_w[i] * invstd;
  p.aff[((size_t)b * p.nsub + band) * p.cout + ch] =
      make_float2((float)(g * sc), (float)(g * ((double)p.bn_b[i] - sc * mean) + be));
}

struct ApplyParams {
  float* yraw;             // layer 2: the slots of rows past raw_rows / columns past raw_cols are zeroed on the way (the strip kernel
                           // writes valid positions only; the backward pass reads every slot and 0 * garbage must stay 0)
  int raw_rows;
  const float2* aff;
  float* out;
  const unsigned char* mask;   // optional Dropout keep-mask over the pooled output (layer 1), same layout as `out`
  float mask_scale;            // 1 / (1 - p)
  int B, nsub, tiles_r, tiles_c, out_rows, out_cols;
  long long units;   // B * nsub * tiles_r * tiles_c * NT * 64
  // f16 training (layer 1): the pooled, Dropout-masked activation once more as f16, channel-minor
  // [clip][band][row][col][32], times the (clip, band) range scale -- the operand layout of conv2_f16x3_kernel
  _Float16* out_h16;
  const float* f16_scale;   // [B][nsub][2] = (s, 1/s)
  int raw_cols;             // layer 2: valid columns of the convolution output plane
  _Float16* out_l16;        // split-precision training forward: the low part (v * s - hi) next to out_h16
  int pool_h;               // layer 1 on 2 x 40 tiles: 1 = MaxPool2d((1, 5)) (16-mel sub-bands: two 1 x 5 windows per 2 x 5 block), else (2, 5)
  const float* y_scale;     // non-NULL: yraw holds float16 times y_scale[band][0] (ConvParams::yraw16)
  // Dropout drawn IN the kernel (layer 1): element o is kept iff philox(seed, o) >= drop_thresh (= p * 2^32); the keep-mask is
  // written to mask_out (same layout as `out`) for the backward pass.  mask_out == NULL: the caller's `mask` (or none).
  unsigned char* mask_out;
  unsigned long long seed;
  unsigned drop_thresh;
};

using mst::dropout_keep;   // Philox-2x32-10 keyed dropout decisions (common.h)

template <int LAYER, int SUB>
__global__ __launch_bounds__(256) void apply_kernel(const ApplyParams p) {
  using C = CC<LAYER, SUB>;
  constexpr int NT = C::NT, NV = 4 * C::MT;
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  if (u >= p.units) return;
  const int lane = (int)(u & 63), n = (int)((u >> 6) % NT);
  long long tile = (u >> 6) / NT;
  const int tc = (int)(tile % p.tiles_c);
  tile /= p.tiles_c;
  const int tr = (int)(tile % p.tiles_r);
  tile /= p.tiles_r;
  const int band = (int)(tile % p.nsub), clip = (int)(tile / p.nsub);
  const int j = lane & 15, g = lane >> 4, ch = n * 16 + j;
  const float2 ac = p.aff[((size_t)clip * p.nsub + band) * C::COUT + ch];
  float v[NV];
  load_unit<NV>(p.yraw, p.y_scale, band, (size_t)u, v);
  if constexpr (LAYER == 2) {   // slots that are no positions of the plane: keep them 0 for the backward pass (it reads every slot)
    bool touched = false;
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
      const bool no_row = 8 * tr + 4 * (g >> 1) + t >= p.raw_rows;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (no_row || 8 * tc + 4 * (g & 1) + r >= p.raw_cols) v[4 * t + r] = 0.f, touched = true;
    }
    if (touched) {
#pragma unroll
      for (int t = 0; t < C::MT; ++t) {
        if (p.y_scale) {
          h16x4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (_Float16)(v[4 * t + r] * p.y_scale[band * 2]);   // exact: v came from h / s
          reinterpret_cast<h16x4*>(p.yraw)[yvec((size_t)u, t, C::MT)] = h;
        } else {
          reinterpret_cast<f32x4*>(p.yraw)[yvec((size_t)u, t, C::MT)] = f32x4{v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]};
        }
      }
    }
  }
  if constexpr (LAYER == 1) {
    const int nh = (SUB == 2 && p.pool_h == 1) ? 2 : 1;   // pool height 1 on the 2-row tiles: one window per tile row
#pragma unroll
    for (int wv = 0; wv < C::WPG; ++wv)
#pragma unroll
    for (int hr = 0; hr < (SUB == 2 ? 2 : 1); ++hr) {
      if (hr >= nh) continue;
      float m = 0.f;
#pragma unroll
      for (int pos = 0; pos < C::WIN; ++pos) {
        const bool mine = nh == 1 || pos / 5 == hr;
        m = mine ? fmaxf(m, fmaf(v[wv * C::WIN + pos], ac.x, ac.y)) : m;
      }
      m = (ac.x + ac.y == ac.x + ac.y) ? m : ac.x + ac.y;   // poisoned batch statistics (NaN affine): fmaxf would drop the NaN and hand on zeros
      const int pc = 4 * C::WPG * tc + C::WPG * g + wv;
      const int prow = nh == 1 ? tr : 2 * tr + hr;
      if (pc < p.out_cols && prow < p.out_rows) {
        const size_t o = ((((size_t)clip * p.nsub + band) * C::COUT + ch) * p.out_rows + prow) * p.out_cols + pc;
        bool keep = true;
        if (p.mask_out) {   // kernel-uniform: draw here, remember for the backward pass
          keep = dropout_keep(p.seed, o, p.drop_thresh);
          p.mask_out[o] = keep ? 1 : 0;
        } else if (p.mask) {
          keep = p.mask[o] != 0;
        }
        const float v = (p.mask_out || p.mask) ? (keep ? m * p.mask_scale : 0.f) : m;
        if (p.out) p.out[o] = v;   // (NULL in the float16 modes unless the caller asked for the fp32 tensor)
        if (p.out_h16) {
          const size_t o16 = ((((size_t)clip * p.nsub + band) * p.out_rows + prow) * p.out_cols + pc) * 32 + ch;
          const float vs = v * p.f16_scale[((size_t)clip * p.nsub + band) * 2];
          const _Float16 h = (_Float16)vs;
          p.out_h16[o16] = h;
          if (p.out_l16) p.out_l16[o16] = (_Float16)(vs - (float)h);
        }
      }
    }
  } else {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) m = fmaxf(m, fmaf(v[e], ac.x, ac.y));
    m = (ac.x + ac.y == ac.x + ac.y) ? m : ac.x + ac.y;   // poisoned batch statistics: see layer 1
    const int pr = 2 * tr + (g >> 1), pc = 2 * tc + (g & 1);
    if (pr < p.out_rows && pc < p.out_cols)
      p.out[(((size_t)clip * p.nsub + band) * C::COUT + ch) * p.out_rows * p.out_cols + (size_t)pr * p.out_cols + pc] = m;
  }
}

// ---- backward of everything between the raw convolution output y and the pooled activation:
// max-pool -> ReLU -> FiLM -> BatchNorm(batch statistics), recomputed per lane from yraw.
//   f = A y + C (A, C as in the forward), window max / arg-max (first maximum in row-major order, as PyTorch), ReLU;
//   df = dpool at the arg-max if the max is positive;  dgamma_film += df z, dbeta_film += df  (z = BN output);
//   dz = gamma_film df;  S1 = sum dz (= dbeta_bn),  S2 = sum dz zhat (= dgamma_bn)  over the batch;
//   dy = gamma_bn / std * (dz - S1/N - zhat S2/N)   at every valid position (also where dz = 0).
// Pass A (reduce) accumulates the sums, pass B (dx) writes dy as [band][B][COUT][rows][cols] (NCHW per band: the
// operand layout of the convolution backward that consumes it).
struct ApplyBwdParams {
  const float* yraw;
  const float2* aff;       // [B][nsub][COUT]
  const float2* bnstat;    // [nsub][COUT] (mean, 1/std)
  const float* bn_w;       // [nsub][COUT]
  const float* bn_b;
  const float* film;       // [B][nsub*192]
  const float* dpool;      // upstream gradient of the pooled activation
  long long dp_clip, dp_band, dp_ch;   // its strides (floats); rows are dp_cols apart
  int dp_rows, dp_cols;
  float* dfilm;            // [B][nsub*192]  (+=, by dfilm_finish_kernel from dfilm_acc)
  mst::DetAcc* dfilm_acc;  // [B][nsub*192] order-independent accumulators of this call (zeroed per call)
  mst::DetAcc* sums;       // [nsub][COUT][2] (+=)  S1, S2
  float* dy;               // pass B: NCHW per band, or NULL: in place over yraw in accumulator order (dy_acc)
  float* dy_acc;
  // f16 training: dpool is multiplied by in_scale[0] when it is read (layer 2: the backward pass's internal loss scale;
  // NULL = 1), and layer 2's dy goes out as f16, channel-minor [band][clip][row][col][64] (conv2_dgrad_f16_kernel's operand)
  const float* in_scale;
  _Float16* dy_h16;
  int pool_h;              // layer 1 on 2 x 40 tiles: 1 = pooling windows of 1 x 5 (see ApplyParams::pool_h)
  const float* y_scale;    // non-NULL: yraw holds float16 times y_scale[band][0] (f16 training)
  int B, nsub, tiles_r, tiles_c, rows, cols, goff, boff;
  double count;
  int chunks;              // pass A: blocks per (clip, band)
  const long long* clips;  // cross-rank statistics: see FoldParams::clips (count = clips[0] * rows * cols)
};
__device__ __forceinline__ double bwd_count(const ApplyBwdParams& p) {
  return p.clips ? (double)p.clips[0] * (double)p.rows * (double)p.cols : p.count;
}
__global__ void set_clips_kernel(long long* a, long long* b, long long clips) {
  if (a) a[0] = clips, a[1] = 0;
  if (b) b[0] = clips, b[1] = 0;
}

template <int LAYER, int SUB>
__device__ __forceinline__ void unit_geometry(int tr, int tc, int g, int e, int& row, int& col) {
  using C = CC<LAYER, SUB>;
  if constexpr (LAYER == 1) {
    const int wv = e / C::WIN, pos = e % C::WIN;
    row = C::TROWS * tr + pos / 5;
    col = C::TCOLS * tc + 5 * (C::WPG * g + wv) + pos % 5;
  } else {
    row = 8 * tr + 4 * (g >> 1) + (e >> 2);
    col = 8 * tc + 4 * (g & 1) + (e & 3);
  }
}

// df for the NV values of one lane-unit: zero except at the arg-max of every pooling window with a positive maximum
template <int LAYER, int SUB>
__device__ __forceinline__ void unit_df(const ApplyBwdParams& p, const float (&v)[4 * CC<LAYER, SUB>::MT], float2 ac, int clip,
                                        int band, int ch, int tr, int tc, int g, float (&df)[4 * CC<LAYER, SUB>::MT]) {
  using C = CC<LAYER, SUB>;
  constexpr int NV = 4 * C::MT;
#pragma unroll
  for (int e = 0; e < NV; ++e) df[e] = 0.f;
  const float* dpb = p.dpool + clip * p.dp_clip + band * p.dp_band + ch * p.dp_ch;
  const float sin = p.in_scale ? p.in_scale[0] : 1.0f;
  constexpr int NW = LAYER == 1 ? C::WPG : 1, WIN = LAYER == 1 ? C::WIN : NV;
  if (LAYER == 1 && SUB == 2 && p.pool_h == 1) {   // MaxPool2d((1, 5)) on the 2-row tiles: windows = the 5 columns of ONE tile row
#pragma unroll
    for (int wv = 0; wv < NW; ++wv)
#pragma unroll
      for (int hr = 0; hr < 2; ++hr) {
        float m = 0.f;
        int am = -1;
#pragma unroll
        for (int pos = 0; pos < 5; ++pos) {
          const int e = wv * WIN + 5 * hr + pos;
          const float f = fmaf(v[e], ac.x, ac.y);
          if (f > m) m = f, am = e;
        }
        const int pr = 2 * tr + hr, pc = 4 * C::WPG * tc + C::WPG * g + wv;
        const float d = (am >= 0 && pr < p.dp_rows && pc < p.dp_cols) ? dpb[(size_t)pr * p.dp_cols + pc] * sin : 0.f;
#pragma unroll
        for (int pos = 0; pos < 5; ++pos) {
          const int e = wv * WIN + 5 * hr + pos;
          df[e] = (e == am) ? d : 0.f;
        }
      }
    return;
  }
#pragma unroll
  for (int wv = 0; wv < NW; ++wv) {
    float m = 0.f;
    int am = -1;
#pragma unroll
    for (int pos = 0; pos < WIN; ++pos) {
      const float f = fmaf(v[wv * WIN + pos], ac.x, ac.y);
      if (f > m) m = f, am = wv * WIN + pos;
    }
    int pr, pc;
    if constexpr (LAYER == 1) pr = tr, pc = 4 * C::WPG * tc + C::WPG * g + wv;
    else pr = 2 * tr + (g >> 1), pc = 2 * tc + (g & 1);
    const float d = (am >= 0 && pr < p.dp_rows && pc < p.dp_cols) ? dpb[(size_t)pr * p.dp_cols + pc] * sin : 0.f;
#pragma unroll
    for (int e = 0; e < WIN; ++e) df[wv * WIN + e] = (wv * WIN + e == am) ? d : 0.f;
  }
}

template <int LAYER, int SUB>
__global__ __launch_bounds__(256) void apply_bwd_reduce_kernel(const ApplyBwdParams p) {   // grid (chunks, B*nsub)
  using C = CC<LAYER, SUB>;
  constexpr int NT = C::NT, NV = 4 * C::MT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int clip = blockIdx.y / p.nsub, band = blockIdx.y % p.nsub;
  const int wus = p.tiles_r * p.tiles_c * NT;                 // wave-units of this (clip, band)
  const int per = ((wus + p.chunks - 1) / p.chunks + 3) & ~3;  // multiple of 4: a wave keeps its N-tile
  const int w0 = blockIdx.x * per, w1 = min(wus, w0 + per);
  const int j = lane & 15, g = lane >> 4;
  float dgam = 0.f, dbet = 0.f;
  double s1 = 0.0, s2 = 0.0;
  int ch = 0;
  for (int wu = w0 + wave; wu < w1; wu += 4) {
    const int n = wu % NT, tile = wu / NT, tc = tile % p.tiles_c, tr = tile / p.tiles_c;
    ch = n * 16 + j;
    const float2 ac = p.aff[((size_t)clip * p.nsub + band) * C::COUT + ch];
    const float2 ms = p.bnstat[band * C::COUT + ch];
    const float gb = p.bn_w[band * C::COUT + ch], bb = p.bn_b[band * C::COUT + ch];
    const float gf = p.film[((size_t)clip * p.nsub + band) * 192 + p.goff + ch];
    const size_t u = ((((size_t)clip * p.nsub + band) * p.tiles_r + tr) * p.tiles_c + tc) * NT + n;
    float v[NV], df[NV];
    load_unit<NV>(p.yraw, p.y_scale, band, u * 64 + lane, v);
    unit_df<LAYER, SUB>(p, v, ac, clip, band, ch, tr, tc, g, df);
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      int row, col;
      unit_geometry<LAYER, SUB>(tr, tc, g, e, row, col);
      const float zh = (row < p.rows && col < p.cols) ? (v[e] - ms.x) * ms.y : 0.f;   // padding slots: df is 0 there, keep 0 * x finite
      dgam = fmaf(df[e], fmaf(gb, zh, bb), dgam);
      dbet += df[e];
      const double dz = (double)gf * (double)df[e];
      s1 += dz, s2 += dz * (double)zh;
    }
  }
  // lanes j, j+16, j+32, j+48 share the channel
  dgam += __shfl_xor(dgam, 16, 64), dgam += __shfl_xor(dgam, 32, 64);
  dbet += __shfl_xor(dbet, 16, 64), dbet += __shfl_xor(dbet, 32, 64);
  s1 += __shfl_xor(s1, 16, 64), s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 16, 64), s2 += __shfl_xor(s2, 32, 64);
  if (lane < 16 && w0 + wave < w1) {
    mst::DetAcc* fl = p.dfilm_acc + ((size_t)clip * p.nsub + band) * 192;
    mst::det_add(fl + p.goff + ch, (double)dgam);
    mst::det_add(fl + p.boff + ch, (double)dbet);
    mst::det_add(p.sums + ((size_t)band * C::COUT + ch) * 2, s1);
    mst::det_add(p.sums + ((size_t)band * C::COUT + ch) * 2 + 1, s2);
  }
}

template <int LAYER, int SUB>
__global__ __launch_bounds__(256) void apply_bwd_dx_kernel(const ApplyBwdParams p, long long units) {
  using C = CC<LAYER, SUB>;
  constexpr int NT = C::NT, NV = 4 * C::MT;
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  const int lane = (int)(u & 63), n = (int)((u >> 6) % NT);
  long long tile = (u >> 6) / NT;
  const int tc = (int)(tile % p.tiles_c);
  tile /= p.tiles_c;
  const int tr = (int)(tile % p.tiles_r);
  tile /= p.tiles_r;
  const int band = (int)(tile % p.nsub), clip = (int)(tile / p.nsub);
  const int j = lane & 15, g = lane >> 4, ch = n * 16 + j;
  const float2 ac = p.aff[((size_t)clip * p.nsub + band) * C::COUT + ch];
  const float2 ms = p.bnstat[band * C::COUT + ch];
  const float gb = p.bn_w[band * C::COUT + ch];
  const float gf = p.film[((size_t)clip * p.nsub + band) * 192 + p.goff + ch];
  const double cnt = bwd_count(p);
  const double m1 = mst::det_get(p.sums[((size_t)band * C::COUT + ch) * 2]) / cnt,
               m2 = mst::det_get(p.sums[((size_t)band * C::COUT + ch) * 2 + 1]) / cnt;
  const f32x4* src = reinterpret_cast<const f32x4*>(p.yraw);
  float v[NV], df[NV];
#pragma unroll
  for (int t = 0; t < C::MT; ++t) {
    const f32x4 q = src[yvec((size_t)u, t, C::MT)];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[4 * t + r] = q[r];
  }
  unit_df<LAYER, SUB>(p, v, ac, clip, band, ch, tr, tc, g, df);
  const float k = gb * ms.y;
  if (p.dy_acc != nullptr) {   // in place, accumulator order (operand layout of the hand-written weight gradient); 0 outside
    f32x4* dst = reinterpret_cast<f32x4*>(p.dy_acc);
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
      f32x4 q;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = 4 * t + r;
        int row, col;
        unit_geometry<LAYER, SUB>(tr, tc, g, e, row, col);
        const float zh = (v[e] - ms.x) * ms.y;
        q[r] = (row < p.rows && col < p.cols) ? k * (float)((double)gf * (double)df[e] - m1 - (double)zh * m2) : 0.f;
      }
      dst[yvec((size_t)u, t, C::MT)] = q;
    }
  }
  if (p.dy_h16 != nullptr) {
    _Float16* dyh = p.dy_h16 + ((size_t)band * p.B + clip) * (size_t)p.rows * p.cols * C::COUT + ch;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      int row, col;
      unit_geometry<LAYER, SUB>(tr, tc, g, e, row, col);
      if (row < p.rows && col < p.cols) {
        const float zh = (v[e] - ms.x) * ms.y;
        dyh[((size_t)row * p.cols + col) * C::COUT] = to_f16_sat(k * (float)((double)gf * (double)df[e] - m1 - (double)zh * m2));
      }
    }
  }
  if (p.dy == nullptr) return;
  float* dyb = p.dy + (((size_t)band * p.B + clip) * C::COUT + ch) * (size_t)p.rows * p.cols;
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    int row, col;
    unit_geometry<LAYER, SUB>(tr, tc, g, e, row, col);
    if (row < p.rows && col < p.cols) {
      const float zh = (v[e] - ms.x) * ms.y;
      dyb[(size_t)row * p.cols + col] = k * (float)((double)gf * (double)df[e] - m1 - (double)zh * m2);
    }
  }
}

// ------------------------------------------------------------------------------------------
// conv1 weight gradient:  dW[band][co][ci][tap] = sum over clips and positions of dy[co][pos] * x[ci][pos + tap]
// as an fp32-MFMA GEMM with K = positions.  dy arrives in ACCUMULATOR ORDER (apply_bwd_dx_kernel, in place over yraw):
// register e of lane (j, g) is dy[co = 16 n + j][position P(g, e)] -- exactly the A operand A[i = co][k = g] of k-step e.
// The B operand B[k = g][n] = x[ci_n][P(g, e) + tap_n] is one ds_read_b32 from the same haloed 8-channel patch the
// forward kernel stages (address = per-N-tile lane base + 10 g + a compile-time offset of e).
// All 8 waves of a workgroup work on the SAME position tile and own different N-tiles of the 392 (ci, tap) columns
// (25 tiles of 16, dealt round-robin), so a wave keeps 2 x <=4 accumulator tiles for the workgroup's whole contiguous
// run of tiles and adds them to dW with atomics when the band changes / at the end.
// ------------------------------------------------------------------------------------------
struct WgradParams {
  const float* x;          // logmel [B][8][n_mels][frames]
  const float* dy;         // accumulator order [B][nsub][tr][tc][2][64][20]
  mst::DetAcc* dw;         // [nsub][32][8][49] order-independent accumulators (+=, zeroed by the caller; det_to_float_kernel)
  int B, nsub, tiles_r, tiles_c;
  int in_rows, in_cols, in_cstride, in_bandoff;
  long long in_clipstride;
  int dbg;                 // unused (was: timing experiments of the conv2 kernel)
};

// NW waves per workgroup (4: two independent workgroups per CU, so that one's loads / barrier overlap the other's MFMAs)
// SUB: pool height of the layer (tile = SUB rows x 40 / 80 columns, patch 8 x (SUB + 6) x 46 / 86)
template <int SUB, int NW>
__global__ __launch_bounds__(NW * 64, SUB == 2 ? 4 : 2) void conv1_wgrad_kernel(const WgradParams p) {   // SUB 2: two workgroups per CU
  using C = CC<1, SUB>;
  constexpr int PR = C::PR, PC = C::PC, CHS = PR * PC, PATCH = 8 * CHS;
  constexpr int NCV = (PC + 63) / 64;  // 64-lane loads per patch row
  constexpr int GSTEP = 5 * C::WPG;    // column distance between the lane groups' positions
  constexpr int NTN = 25;             // N-tiles of 16 over the 392 (ci, tap) columns
  constexpr int KF = (NTN - 1) / NW;        // full N-tiles per wave: nt = wave + NW*k, k < KF (24 of the 25)
  constexpr bool SHARED = true;             // N-tile 24 (half empty): its 20 k-steps are dealt over the waves
  constexpr int KN = KF + 1;
  constexpr int RPW = (8 * PR * NCV + NW - 1) / NW;   // row pieces (64-lane loads) a wave issues per tile
  static_assert(RPW <= 32, "mask bits");
  __shared__ float patch[2][PATCH];
  __shared__ __attribute__((aligned(16))) float dybuf[2][2 * 64 * 20];   // the tile's dy, fetched ONCE per workgroup
  constexpr int NDY = (2 * 64 * 5 + NW * 64 - 1) / (NW * 64);               // float4 loads per thread
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int G = gridDim.x, wg = mst::xcd_remap(blockIdx.x, G);
  const int tpb = p.B * p.tiles_r * p.tiles_c;            // tiles per band
  const int total = p.nsub * tpb;                          // < 2^31 (checked by the host)
  const int s_begin = (int)((long long)wg * total / G), s_end = (int)((long long)(wg + 1) * total / G);

  // per N-tile lane constants
  int nbase[KN];
  float nmask[KN];
#pragma unroll
  for (int k = 0; k < KF; ++k) {
    const int nidx = (wave + NW * k) * 16 + j;
    const bool ok = true;   // N-tiles 0..23 are full
    const int ci = ok ? nidx / 49 : 0, tap = ok ? nidx % 49 : 0;
    nbase[k] = ci * CHS + (tap / 7) * PC + tap % 7 + GSTEP * g;
    nmask[k] = ok ? 1.f : 0.f;
  }
  // shared N-tile 24: waves 0..3 take 3 k-steps, waves 4..7 two (NW = 8); with NW = 4 every wave takes 5
  const int sh_cnt = NW == 8 ? (wave < 4 ? 3 : 2) : 5;
  const int sh_e0 = NW == 8 ? (wave < 4 ? 3 * wave : 12 + 2 * (wave - 4)) : 5 * wave;
  const int sh_nidx = (NTN - 1) * 16 + j;
  const int sh_base = (sh_nidx < 392 ? (sh_nidx / 49) * CHS + ((sh_nidx % 49) / 7) * PC + (sh_nidx % 49) % 7 : 0) + GSTEP * g;
  const float sh_mask = sh_nidx < 392 ? 1.f : 0.f;

  struct TileId {
    int band, clip, tr, tc;
  };
  auto decode = [&](int s) {   // only once per workgroup: the loop below advances tile ids incrementally (scalar unit)
    TileId t;
    t.band = s / tpb;
    int r = s - t.band * tpb;
    t.clip = r / (p.tiles_r * p.tiles_c);
    r -= t.clip * p.tiles_r * p.tiles_c;
    t.tr = r / p.tiles_c, t.tc = r - t.tr * p.tiles_c;
    return t;
  };
  auto advance = [&](TileId t) {
    if (++t.tc == p.tiles_c) {
      t.tc = 0;
      if (++t.tr == p.tiles_r) {
        t.tr = 0;
        if (++t.clip == p.B) t.clip = 0, ++t.band;
      }
    }
    return t;
  };
  // patch rows of a tile: row = cc * PR + r -> 8 per wave, one 64-lane load each
  // prefetch of a tile in RPW + NDY pieces (one vector-memory instruction each), issued BETWEEN the MFMAs of the
  // current tile: all waves of the workgroup run in lockstep (one barrier per tile), so a block of loads at the top
  // of the tile would leave every matrix pipe idle while it issues
  const float* pf_src = p.x;
  int pf_row0 = 0, pf_col0 = 0;
  const f32x4* dy_src = reinterpret_cast<const f32x4*>(p.dy);
  auto prefetch_setup = [&](const TileId& t) __attribute__((always_inline)) {
    pf_src = p.x + (size_t)t.clip * p.in_clipstride + (size_t)t.band * p.in_bandoff;
    pf_row0 = C::TROWS * t.tr - 3;
    pf_col0 = C::TCOLS * t.tc - 3;
    const size_t u = ((((size_t)t.clip * p.nsub + t.band) * p.tiles_r + t.tr) * p.tiles_c + t.tc) * 2;
    dy_src = reinterpret_cast<const f32x4*>(p.dy + u * 64 * 20);
  };
  auto prefetch_piece = [&](int i, float (&pf)[RPW], f32x4 (&dq)[NDY], unsigned& pmask) __attribute__((always_inline)) {
    if (i < RPW) {
      const int idx = wave * RPW + i, row = min(idx / NCV, 8 * PR - 1), cv = idx % NCV;
      const int cc = row / PR, r = row % PR, col = lane + 64 * cv;
      const int rin = pf_row0 + r, rc = min(max(rin, 0), p.in_rows - 1);
      const int cin = pf_col0 + col, cl = min(max(cin, 0), p.in_cols - 1);
      // the out-of-image mask is applied when the row is staged: touching the value here would make the wave wait
      // for the load on the spot
      pf[i] = (pf_src + (size_t)cc * p.in_cstride + (size_t)rc * p.in_cols)[cl];
      if (idx < 8 * PR * NCV && col < PC && cin == cl && rin == rc) pmask |= 1u << i;
    } else if (i < RPW + NDY) {
      dq[i - RPW] = dy_src[min(tid + (i - RPW) * NW * 64, 2 * 64 * 5 - 1)];
    }
  };
  auto prefetch = [&](const TileId& t, float (&pf)[RPW], f32x4 (&dq)[NDY], unsigned& pmask) __attribute__((always_inline)) {
    prefetch_setup(t);
    pmask = 0;
#pragma unroll
    for (int i = 0; i < RPW + NDY; ++i) prefetch_piece(i, pf, dq, pmask);
  };
  auto stage = [&](int buf, const float (&pf)[RPW], unsigned pmask) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int idx = wave * RPW + i, row = idx / NCV, col = lane + 64 * (idx % NCV);
      if (idx < 8 * PR * NCV && col < PC) patch[buf][row * PC + col] = ((pmask >> i) & 1u) ? pf[i] : 0.f;
    }
  };
  f32x4 acc[2][KN];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int k = 0; k < KN; ++k) acc[c][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto flush = [&](int band) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int nt = k < KF ? wave + NW * k : NTN - 1;
      const int nidx = nt * 16 + j;
      if (nidx < 392) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = 16 * c + 4 * g + r;
            mst::det_add(p.dw + ((size_t)band * 32 + co) * 392 + nidx, (double)acc[c][k][r]);
            acc[c][k][r] = 0.f;
          }
      }
    }
  };

  if (s_begin >= s_end) return;
  int cur_band = -1;
  auto stage_dy = [&](int buf, const f32x4 (&dq)[NDY]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NDY; ++i)
      if (tid + i * NW * 64 < 2 * 64 * 5) reinterpret_cast<f32x4*>(dybuf[buf])[tid + i * NW * 64] = dq[i];
  };
  // global -> register prefetch runs TWO tiles ahead (two explicit register sets, loop unrolled by two): the data of
  // tile s+1 is staged into LDS at the end of tile s, a full tile after its loads were issued
  float pfA[RPW], pfB[RPW];
  f32x4 dqA[NDY], dqB[NDY];
  unsigned pmA = 0, pmB = 0;
  TileId cur = decode(s_begin), nxt = cur, nxt2 = cur;
  prefetch(cur, pfA, dqA, pmA);
  stage(0, pfA, pmA);
  stage_dy(0, dqA);
  if (s_begin + 1 < s_end) nxt2 = advance(cur);
  prefetch(nxt2, pfA, dqA, pmA);      // tile s_begin + 1
  __syncthreads();
  auto body = [&](int s, int buf, float (&pfs)[RPW], f32x4 (&dqs)[NDY], unsigned& pms, float (&pfn)[RPW], f32x4 (&dqn)[NDY],
                  unsigned& pmn) __attribute__((always_inline)) {
    // on entry: nxt2 = tile s+1 (its data sits in pfs / dqs); this call fetches tile s+2 into pfn / dqn
    cur = nxt;
    nxt = nxt2;
    if (s + 2 < s_end) nxt2 = advance(nxt);   // past the run: re-fetch the last tile (never used)
    const int band = cur.band;
    if (band != cur_band) {
      if (cur_band >= 0) flush(cur_band);
      cur_band = band;
    }
    prefetch_setup(nxt2);
    pmn = 0;
    __builtin_amdgcn_sched_barrier(0);
    const float* pb = patch[buf];
    const float* dyl = dybuf[buf];
    // A operands from the shared copy, four k-steps (one 16-byte read per M-tile) at a time into a double buffer, one group ahead of
    // their MFMAs (pitch 5 units per lane: conflict-free).  Holding the whole tile's 2 x 5 vectors cost 24 more registers and kept
    // the kernel at one workgroup per CU -- two waves per SIMD in lockstep, nobody to cover a barrier or an LDS round trip.
    f32x4 ag[2][2];
    auto read_a = [&](int grp) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < 2; ++c) ag[grp & 1][c] = reinterpret_cast<const f32x4*>(dyl)[(c * 5 + grp % 5) * 64 + lane];   // t-major copy
    };
    read_a(0);
    // B operands: one float per k-step, read from LDS into a ring of 4 THREE k-steps ahead, one read behind
    // the first MFMA of every step (pinned by sched_barrier).  Before, a block of 20 reads stood in front of every N-tile, and the
    // two waves of a SIMD -- in lockstep, one barrier per tile -- issued theirs together while the matrix pipe idled.
    constexpr int NS = KF * 20, LA = 3;
    float bq[4];
    auto read_b = [&](int s2) __attribute__((always_inline)) {
      const int k2 = s2 / 20, e2 = s2 % 20, wv = e2 / C::WIN, pos = e2 % C::WIN;
      bq[s2 & 3] = pb[nbase[k2] + (pos / 5) * PC + 5 * wv + pos % 5];
    };
#pragma unroll
    for (int s2 = 0; s2 < LA; ++s2) read_b(s2);
    int piece = 0;
#pragma unroll
    for (int s1 = 0; s1 < NS; ++s1) {   // the wave's own full N-tiles, flat step s1 = 20 k + e
      const int k = s1 / 20, e = s1 % 20, grp = s1 / 4;
      const float b = bq[s1 & 3];
      acc[0][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[grp & 1][0][e & 3], b, acc[0][k], 0, 0, 0);
      if (s1 + LA < NS) read_b(s1 + LA);
      if ((s1 & 3) == 0 && s1 + 4 < NS) read_a(grp + 1);
      __builtin_amdgcn_sched_barrier(0);
      acc[1][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[grp & 1][1][e & 3], b, acc[1][k], 0, 0, 0);
      // one prefetch instruction per 10 MFMAs (12 slots), or per 5 when the tile needs more than 12 pieces
      if ((RPW + NDY > 12 ? ((e % 5) == 1 || (e % 5) == 3) : (e % 5) == 2)) {
        prefetch_piece(piece, pfn, dqn, pmn);
        ++piece;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SHARED) {   // the last, shared N-tile: this wave's share of its 20 k-steps
#pragma unroll
      for (int q = 0; q < (NW == 8 ? 3 : 5); ++q) {
        if (q < sh_cnt) {   // wave-uniform
          const int e = sh_e0 + q;
          const int wv = e / C::WIN, pos = e % C::WIN;
          const float b = pb[sh_base + (pos / 5) * PC + 5 * wv + pos % 5] * sh_mask;
          const float a0 = dyl[((0 * 5 + (e >> 2)) * 64 + lane) * 4 + (e & 3)], a1 = dyl[((1 * 5 + (e >> 2)) * 64 + lane) * 4 + (e & 3)];
          acc[0][KF] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0][KF], 0, 0, 0);
          acc[1][KF] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1][KF], 0, 0, 0);
        }
      }
    }
    stage(buf ^ 1, pfs, pms);
    stage_dy(buf ^ 1, dqs);
    __syncthreads();
  };
  for (int s = s_begin; s < s_end; s += 2) {
    body(s, 0, pfA, dqA, pmA, pfB, dqB, pmB);
    if (s + 1 < s_end) body(s + 1, 1, pfB, dqB, pmB, pfA, dqA, pmA);
  }
  flush(cur_band);
}

// ------------------------------------------------------------------------------------------
// conv2 weight gradient, same scheme as conv1_wgrad_kernel: dW2[band][co][ci][tap] with M = 64 output channels (4 tiles),
// the 32 x 49 (ci, tap) columns in 4 chunks of 8 input channels (392 columns = 24 full N-tiles + a shared half tile,
// exactly conv1's N structure), K = positions: 8 x 8 tiles of the 10 x 344 plane, dy2 in accumulator order (16 values
// per lane: position (4 (g >> 1) + e / 4, 4 (g & 1) + e % 4)).  Tiles of the second tile row hold output rows 8 and 9
// only: their k-steps e >= 8 are all zero and are skipped.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kConvThreads) void conv2_wgrad_kernel(const WgradParams p) {
  constexpr int NW = 8, PR = 14, PC = 14, CHS = PR * PC, PATCH = 8 * CHS;   // 8 channels x 14 x 14
  constexpr int NTN = 25, KF = 3, KN = KF + 1, NCH = 4;
  constexpr int NPF = 4;        // patch elements per thread: flat index f = tid + 512 i over [8*14 rows][16 pitch]
  constexpr int NDY = 2;        // float4 of dy per thread: 4 * 64 * 16 floats per tile
  __shared__ float patch[2][PATCH];
  // LDS copy of the tile's dy: a linear copy of the t-major tensor, [M-tile][vector q][lane] -- consecutive lanes read
  // consecutive 16-byte units, conflict-free (the lane-major copy of the first version needed a padded pitch of 5 units)
  __shared__ __attribute__((aligned(16))) float dybuf[2][4 * 64 * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int G = gridDim.x, wg = mst::xcd_remap(blockIdx.x, G);
  // work = (band, tile) pairs, dealt in contiguous ranges to GROUPS of 4 workgroups; the 4 members of a group walk the
  // same range, one input-channel chunk each, side by side on one XCD: the tile's dy is fetched from HBM once and
  // found in L2 by the other three
  const int tpc = p.B * p.tiles_r * p.tiles_c;           // tiles per band
  const int total = p.nsub * tpc;
  const int my_chunk = wg & (NCH - 1), rg = wg >> 2, RG = G >> 2;
  const int s_begin = (int)((long long)rg * total / RG), s_end = (int)((long long)(rg + 1) * total / RG);
  if (s_begin >= s_end) return;

  const int goff = (4 * (g >> 1)) * PC + 4 * (g & 1);
  int nbase[KF];
#pragma unroll
  for (int k = 0; k < KF; ++k) {
    const int nidx = (wave + NW * k) * 16 + j, ci = nidx / 49, tap = nidx % 49;
    nbase[k] = ci * CHS + (tap / 7) * PC + tap % 7 + goff;
  }
  const int sh_nidx = (NTN - 1) * 16 + j;
  const int sh_base = (sh_nidx < 392 ? (sh_nidx / 49) * CHS + ((sh_nidx % 49) / 7) * PC + (sh_nidx % 49) % 7 : 0) + goff;
  const float sh_mask = sh_nidx < 392 ? 1.f : 0.f;
  // LOW tiles (the last tile row when it holds at most 2 output rows: rows 8, 9 of 10 at the default geometry).  In the forward
  // pass's accumulator order their 16 positions sit in slots 0..7 of the two upper lane groups and the two lower groups are
  // padding, i.e. half of every MFMA's K.  Here the lower groups take the tile's SECOND row instead: lane group g works on
  // (row g >> 1, columns 4 (g & 1) .. + 3) -- its A operand is slot 4 + e of the lane 32 below it, its B operand the patch
  // one row down -- and the tile is done in 4 k-steps instead of 8 (conv2's weight gradient: 24 -> 20 k-steps per tile column).
  const int glow = (g >> 1) * PC + 4 * (g & 1) - goff;   // added to nbase / sh_base in the low tiles

  struct Item {
    int band, chunk, clip, tr, tc;
  };
  auto decode = [&](int s) __attribute__((always_inline)) {
    Item t;
    t.band = s / tpc;
    int r = s - t.band * tpc;
    t.chunk = my_chunk;
    t.clip = r / (p.tiles_r * p.tiles_c);
    r -= t.clip * p.tiles_r * p.tiles_c;
    t.tr = r / p.tiles_c, t.tc = r - t.tr * p.tiles_c;
    return t;
  };
  auto advance = [&](Item t) __attribute__((always_inline)) {
    if (++t.tc == p.tiles_c) {
      t.tc = 0;
      if (++t.tr == p.tiles_r) {
        t.tr = 0;
        if (++t.clip == p.B) t.clip = 0, ++t.band;
      }
    }
    return t;
  };

  const float* pf_src = p.x;
  int pf_row0 = 0, pf_col0 = 0;
  const f32x4* dy_src = reinterpret_cast<const f32x4*>(p.dy);
  auto prefetch_setup = [&](const Item& t) __attribute__((always_inline)) {
    pf_src = p.x + (size_t)t.clip * p.in_clipstride + (size_t)t.band * p.in_bandoff + (size_t)(8 * t.chunk) * p.in_cstride;
    pf_row0 = 8 * t.tr - 3, pf_col0 = 8 * t.tc - 3;
    const size_t u = ((((size_t)t.clip * p.nsub + t.band) * p.tiles_r + t.tr) * p.tiles_c + t.tc) * 4;
    dy_src = reinterpret_cast<const f32x4*>(p.dy + u * 64 * 16);
  };
  auto prefetch_piece = [&](int i, float (&pf)[NPF], f32x4 (&dq)[NDY], unsigned& pmask) __attribute__((always_inline)) {
    if (i < NPF) {
      const int f = tid + kConvThreads * i, row = f >> 4, col = f & 15;
      const int cc = min(row / PR, 7), r = row % PR;
      const int rin = pf_row0 + r, cin = pf_col0 + col;
      const int rc = min(max(rin, 0), p.in_rows - 1), cl = min(max(cin, 0), p.in_cols - 1);
      pf[i] = (pf_src + (size_t)cc * p.in_cstride + (size_t)rc * p.in_cols)[cl];
      if (row < 8 * PR && col < PC && rin == rc && cin == cl) pmask |= 1u << i;
    } else if (i < NPF + NDY) {
      dq[i - NPF] = dy_src[tid + (i - NPF) * kConvThreads];
    }
  };
  auto stage = [&](int buf, const float (&pf)[NPF], const f32x4 (&dq)[NDY], unsigned pmask) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int f = tid + kConvThreads * i, row = f >> 4, col = f & 15;
      if (row < 8 * PR && col < PC) patch[buf][row * PC + col] = ((pmask >> i) & 1u) ? pf[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      reinterpret_cast<f32x4*>(dybuf[buf])[tid + i * kConvThreads] = dq[i];
    }
  };

  f32x4 acc[4][KN];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < KN; ++k) acc[c][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto flush = [&](int band, int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int nt = k < KF ? wave + NW * k : NTN - 1;
      const int nidx = nt * 16 + j;
      if (nidx < 392) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = 16 * c + 4 * g + r;
            mst::det_add(p.dw + ((size_t)band * 64 + co) * 1568 + chunk * 392 + nidx, (double)acc[c][k][r]);
            acc[c][k][r] = 0.f;
          }
      }
    }
  };

  float pfA[NPF], pfB[NPF];
  f32x4 dqA[NDY], dqB[NDY];
  unsigned pmA = 0, pmB = 0;
  Item cur = decode(s_begin), nxt = cur, nxt2 = cur;
  int cur_band = -1, cur_chunk = -1;
  prefetch_setup(cur);
#pragma unroll
  for (int i = 0; i < NPF + NDY; ++i) prefetch_piece(i, pfA, dqA, pmA);
  stage(0, pfA, dqA, pmA);
  if (s_begin + 1 < s_end) nxt2 = advance(cur);
  prefetch_setup(nxt2);
  pmA = 0;
#pragma unroll
  for (int i = 0; i < NPF + NDY; ++i) prefetch_piece(i, pfA, dqA, pmA);
  __syncthreads();

  auto body = [&](int s, int buf, float (&pfs)[NPF], f32x4 (&dqs)[NDY], unsigned& pms, float (&pfn)[NPF], f32x4 (&dqn)[NDY],
                  unsigned& pmn) __attribute__((always_inline)) {
    cur = nxt;
    nxt = nxt2;
    if (s + 2 < s_end) nxt2 = advance(nxt);
    if (cur.band != cur_band || cur.chunk != cur_chunk) {
      if (cur_band >= 0) flush(cur_band, cur_chunk);
      cur_band = cur.band, cur_chunk = cur.chunk;
    }
    prefetch_setup(nxt2);
    pmn = 0;
    __builtin_amdgcn_sched_barrier(0);
    const float* pb = patch[buf];
    const float* dyl = dybuf[buf];
    // Explicit software pipeline (as in the forward kernels): between the MFMAs of a k-step the wave issues the LDS reads of LATER
    // steps -- the B operand (one float, ring of 4: three k-steps = 384 matrix cycles ahead) and, once per group of four
    // k-steps, the next group's A operands (4 x 16 bytes per M-tile, one group = 512 cycles ahead) -- and two global prefetch
    // instructions per N-tile.  Before, the A reads sat directly in front of their MFMAs and a block of 16 B reads in front of
    // every N-tile: both waves of a SIMD (they run in lockstep, one barrier per tile) stalled on them together.
    const bool low = 8 * __builtin_amdgcn_readfirstlane(cur.tr) + 2 >= p.in_rows;   // second tile row: only output rows 8, 9 exist -> k-steps 0..7 (wave-uniform)
    auto mma = [&](auto ne_c) __attribute__((always_inline)) {   // NE k-steps per N-tile: 16, or 4 in the low tiles (see glow)
      constexpr int NE = decltype(ne_c)::value, NS = KF * NE;   // flat step s = k * NE + e
      constexpr bool LOW = NE == 4;
      constexpr int LA = 3;                                     // B look-ahead in k-steps
      f32x4 ag[2][4];
      float bq[4];
      auto read_b = [&](int s2) __attribute__((always_inline)) {
        const int k2 = s2 / NE, e2 = s2 % NE;
        bq[s2 & 3] = LOW ? pb[nbase[k2] + glow + e2] : pb[nbase[k2] + (e2 >> 2) * PC + (e2 & 3)];
      };
      auto read_a = [&](int grp) __attribute__((always_inline)) {   // group = 4 k-steps of one N-tile: the same dy values for every N-tile
        const int q = grp % (NE / 4);
#pragma unroll
        for (int c = 0; c < 4; ++c)
          ag[grp & 1][c] = LOW ? reinterpret_cast<const f32x4*>(dyl)[(c * 4 + (lane >> 5)) * 64 + (lane & 31)]
                               : reinterpret_cast<const f32x4*>(dyl)[(c * 4 + q) * 64 + lane];
      };
      read_a(0);
#pragma unroll
      for (int s2 = 0; s2 < LA; ++s2) read_b(s2);
      int piece = 0;
#pragma unroll
      for (int s1 = 0; s1 < NS; ++s1) {
        const int k = s1 / NE, e = s1 % NE, grp = s1 / 4;
        const float b = bq[s1 & 3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[c][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[grp & 1][c][e & 3], b, acc[c][k], 0, 0, 0);
          if (c == 0 && s1 + LA < NS) read_b(s1 + LA);
          if (c == 1 && (e & 3) == 0 && s1 + 4 < NS) read_a(grp + 1);
          if (c == 2 && (LOW ? (e & 1) == 0 : ((e & 3) == 2 && e < 8))) {   // two prefetch instructions per N-tile
            prefetch_piece(piece, pfn, dqn, pmn);
            ++piece;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if (low) mma(std::integral_constant<int, 4>{});
    else mma(std::integral_constant<int, 16>{});
    {   // shared N-tile 24: two k-steps per wave (low tiles: 4 k-steps in all, one each for waves 0..3)
      const int e0 = low ? wave : 2 * wave, cnt = low ? (wave < 4 ? 1 : 0) : 2;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q < cnt) {
          const int e = e0 + q;
          const float b = (low ? pb[sh_base + glow + e] : pb[sh_base + (e >> 2) * PC + (e & 3)]) * sh_mask;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float a = low ? dyl[((c * 4 + (lane >> 5)) * 64 + (lane & 31)) * 4 + e] : dyl[((c * 4 + (e >> 2)) * 64 + lane) * 4 + (e & 3)];
            acc[c][KF] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c][KF], 0, 0, 0);
          }
        }
      }
    }
    stage(buf ^ 1, pfs, dqs, pms);
    __syncthreads();
  };
  for (int s = s_begin; s < s_end; s += 2) {
    body(s, 0, pfA, dqA, pmA, pfB, dqB, pmB);
    if (s + 1 < s_end) body(s + 1, 1, pfB, dqB, pmB, pfA, dqA, pmA);
  }
  flush(cur_band, cur_chunk);
}

#include "encoder_f16train.inc"

// `unscale` (f16 training): (s, 1/s) of the backward pass's internal loss scale; the results are divided by s.  NULL: 1.
__global__ void sums_to_dbn_kernel(const mst::DetAcc* sums, float* dbn, int n, const float* unscale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double u = unscale ? (double)unscale[1] : 1.0;
  if (i < n) dbn[i] = (float)(mst::det_get(sums[2 * i + 1]) * u), dbn[n + i] = (float)(mst::det_get(sums[2 * i]) * u);   // planes: d weight | d bias
}

// dfilm[clip][band][goff + ch] += gamma gradient, [boff + ch] += beta gradient of this layer (single writer per element)
__global__ void dfilm_finish_kernel(const mst::DetAcc* acc, float* dfilm, int n192, int goff, int boff, int cout,
                                    const float* unscale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over B * nsub * 2 * cout
  if (i >= n192 / 192 * 2 * cout) return;
  const int cb = i / (2 * cout), r = i % (2 * cout);
  const int slot = r < cout ? goff + r : boff + (r - cout);
  const double u = unscale ? (double)unscale[1] : 1.0;
  dfilm[(size_t)cb * 192 + slot] += (float)(mst::det_get(acc[(size_t)cb * 192 + slot]) * u);
}

// float result of an order-independent accumulator array (weight gradients)
__global__ void det_to_float_kernel(const mst::DetAcc* acc, float* out, long long n, const float* unscale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const double u = unscale ? (double)unscale[1] : 1.0;
  if (i < n) out[i] = (float)(mst::det_get(acc[i]) * u);
}

// conv weights [nsub][COUT][CIN][49] (device) -> MFMA B-fragment chunks, same layout as conv_fragments() builds on the host
__global__ void conv_fragments_kernel(const float* w, float* f, int nsub, int cout, int cin, int wchp) {
  const int nt = cout / 16, nch = cin / 4;
  const long long total = (long long)nsub * nch * 49 * nt * 64;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  long long r = i >> 6;
  const int n = (int)(r % nt);
  r /= nt;
  const int tap = (int)(r % 49);
  r /= 49;
  const int ch = (int)(r % nch), b = (int)(r / nch);
  const int co = n * 16 + (lane & 15), ci = 4 * ch + (lane >> 4);
  f[((size_t)b * nch + ch) * wchp + ((size_t)tap * nt + n) * 64 + lane] = w[(((size_t)b * cout + co) * cin + ci) * 49 + tap];
}

// conv2 weights [nsub][64 co][32 ci][49] -> B fragments of the input-gradient convolution (64 -> 32, flipped taps):
// f[band][chunk c][tap][nt][lane] = W2[co = 4 c + (lane >> 4)][ci = 16 nt + (lane & 15)][48 - tap]
__global__ void dgrad_fragments_kernel(const float* w, float* f, int nsub, int wchp) {
  const long long total = (long long)nsub * 16 * 49 * 2 * 64;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  long long r = i >> 6;
  const int n = (int)(r % 2);
  r /= 2;
  const int tap = (int)(r % 49);
  r /= 49;
  const int ch = (int)(r % 16), b = (int)(r / 16);
  const int co = 4 * ch + (lane >> 4), ci = n * 16 + (lane & 15);
  f[((size_t)b * 16 + ch) * wchp + ((size_t)tap * 2 + n) * 64 + lane] = w[(((size_t)b * 64 + co) * 32 + ci) * 49 + (48 - tap)];
}

template <int LAYER, int SUB>
hipError_t launch_conv(const ConvParams& cp, int grid, hipStream_t st) {
  using GEO = ConvGeom<LAYER, SUB>;
  const size_t lds = (size_t)(2 * GEO::WBP + kConvWaves * GEO::PATCH) * sizeof(float);
  static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
  if (mst::first_use_on_device(attr_set)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel<LAYER, SUB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((conv_kernel<LAYER, SUB>), dim3(grid), dim3(kConvThreads), lds, st, cp);
  return hipGetLastError();
}

}  // namespace

extern "C" {

int mst_encoder_create(mst_encoder** out, const mst_encoder_config* cfg, const mst_encoder_weights* w) {
  MST_REQUIRE(out, "mst_encoder_create: NULL out");
  *out = nullptr;
  MST_REQUIRE(cfg && w, "mst_encoder_create: NULL config/weights");
  const int ns = cfg->n_subbands;
  MST_REQUIRE(ns >= 1 && cfg->split_size >= 4 && cfg->overlap >= 1 && cfg->n_mels >= cfg->split_size,
              "mst_encoder_create: bad band split (n_mels=%d split=%d overlap=%d n_sub=%d)", cfg->n_mels,
              cfg->split_size, cfg->overlap, ns);
  MST_REQUIRE((ns - 1) * cfg->overlap + cfg->split_size <= cfg->n_mels, "mst_encoder_create: sub-bands exceed n_mels");
  const int sub = cfg->split_size / 10 > 1 ? cfg->split_size / 10 : 1;
  // first-pool heights 1 and 2 run on the MFMA kernels; larger ones on conv1_generic_kernel, whose LDS holds a whole band's patch
  MST_REQUIRE(sub <= 2 || (size_t)(392 * 32 + 8 * (cfg->split_size + 6) * kGenPC) * sizeof(float) <= 160 * 1024,
              "mst_encoder_create: split_size=%d (first-pool height %d) exceeds the generic conv1 kernel's LDS (split_size <= 68)",
              cfg->split_size, sub);
  MST_REQUIRE(cfg->attn_hidden == 256, "mst_encoder_create: attn_hidden must be 256 (got %d)", cfg->attn_hidden);
  MST_REQUIRE(cfg->feature_dim >= 1 && cfg->film_hidden >= 1 && cfg->embed_dim >= 1, "mst_encoder_create: bad dims");
  for (const float* const* q = &w->conv1_w; q <= &w->proj_b; ++q)
    MST_REQUIRE(*q != nullptr, "mst_encoder_create: NULL weight pointer");
  mst_encoder* e = new mst_encoder();
  e->cfg = *cfg;
  e->sub = sub;
  e->H1 = cfg->split_size / sub;
  e->FD = e->H1 / 4;
  e->C = 64 * ns * e->FD;
  MST_REQUIRE(e->FD >= 1 && e->C % 4 == 0, "mst_encoder_create: bad pooled geometry");
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    e->num_cus = prop.multiProcessorCount;
  const int H = cfg->film_hidden, Fd = cfg->feature_dim, A = cfg->attn_hidden, E = cfg->embed_dim, C = e->C;
  // BN(eval) folded with the conv bias, in double
  auto fold = [&](const float* cb, const float* bw, const float* bb, const float* mean, const float* var, int n,
                  std::vector<float>& s, std::vector<float>& t) {
    s.resize(n), t.resize(n);
    for (int i = 0; i < n; ++i) {
      const double sc = (double)bw[i] / std::sqrt((double)var[i] + (double)cfg->bn_eps);
      s[i] = (float)sc;
      t[i] = (float)(sc * ((double)cb[i] - (double)mean[i]) + (double)bb[i]);
    }
  };
  std::vector<float> s1, t1, s2, t2;
  fold(w->conv1_b, w->bn1_w, w->bn1_b, w->bn1_mean, w->bn1_var, ns * 32, s1, t1);
  fold(w->conv2_b, w->bn2_w, w->bn2_b, w->bn2_mean, w->bn2_var, ns * 64, s2, t2);
  auto f1 = conv_fragments(w->conv1_w, ns, 32, 8, ConvGeom<1, 2>::WBP);   // pitch does not depend on SUB
  auto f2 = conv_fragments(w->conv2_w, ns, 64, 32, ConvGeom<2, 2>::WBP);
  auto w0t = transpose(w->mlp0_w, H, Fd), w3t = transpose(w->mlp3_w, H, H), hwt = transpose(w->head_w, ns * 192, H);
  MST_REQUIRE(E % 16 == 0, "mst_encoder_create: embed_dim must be a multiple of 16 (got %d)", E);
  std::vector<float> pfrag((size_t)(C / 4) * (E / 16) * 64);
  for (int s = 0; s < C / 4; ++s)
    for (int n = 0; n < E / 16; ++n)
      for (int lane = 0; lane < 64; ++lane)
        pfrag[((size_t)s * (E / 16) + n) * 64 + lane] = w->proj_w[(size_t)(n * 16 + (lane & 15)) * C + 4 * s + (lane >> 4)];
  std::vector<float> af((size_t)(C / 4) * (A / 16) * 64);
  for (int s = 0; s < C / 4; ++s)
    for (int n = 0; n < A / 16; ++n)
      for (int lane = 0; lane < 64; ++lane)
        af[((size_t)s * (A / 16) + n) * 64 + lane] = w->att0_w[(size_t)(n * 16 + (lane & 15)) * C + 4 * s + (lane >> 4)];
  // per-(band, output channel) power-of-two pre-scale of the f16 weight fragments, and its inverse for the epilogues
  std::vector<int> wex1((size_t)ns * 32), wex2((size_t)ns * 64);
  std::vector<float> winv1((size_t)ns * 32), winv2((size_t)ns * 64);
  for (int b = 0; b < ns; ++b) {
    for (int co = 0; co < 32; ++co) {
      float mx = 0.f;
      for (int i = 0; i < 392; ++i) mx = std::max(mx, fabsf(w->conv1_w[((size_t)b * 32 + co) * 392 + i]));
      wex1[(size_t)b * 32 + co] = f16_weight_exponent(mx);
      winv1[(size_t)b * 32 + co] = ldexpf(1.0f, -wex1[(size_t)b * 32 + co]);
    }
    for (int co = 0; co < 64; ++co) {
      float mx = 0.f;
      for (int i = 0; i < 1568; ++i) mx = std::max(mx, fabsf(w->conv2_w[((size_t)b * 64 + co) * 1568 + i]));
      wex2[(size_t)b * 64 + co] = f16_weight_exponent(mx);
      winv2[(size_t)b * 64 + co] = ldexpf(1.0f, -wex2[(size_t)b * 64 + co]);
    }
  }
  // conv1 weights, f16 hi/lo split of (2^e * w): [band][step][nt][hi/lo][lane][8 channels]
  std::vector<_Float16> f16((size_t)ns * kF16Steps * 2 * 2 * 64 * 8);
  for (int b = 0; b < ns; ++b)
    for (int st = 0; st < kF16Steps; ++st)
      for (int n = 0; n < 2; ++n)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int tap = 4 * st + (lane >> 4), co = n * 16 + (lane & 15);
            const float wv = tap < 49 ? ldexpf(w->conv1_w[(((size_t)b * 32 + co) * 8 + j) * 49 + tap], wex1[(size_t)b * 32 + co]) : 0.f;
            const _Float16 h = (_Float16)wv;
            const size_t base = ((((size_t)b * kF16Steps + st) * 2 + n) * 2) * 64 * 8;
            f16[base + (size_t)lane * 8 + j] = h;
            f16[base + 64 * 8 + (size_t)lane * 8 + j] = (_Float16)(wv - (float)h);
          }
  // ... and in the k order of conv1_f16e_kernel: step = (tap column, upper / lower tap rows), k-group kq -> tap row (0, 2, 1, 3)
  std::vector<_Float16> f16e((size_t)ns * kF16StepsE * 2 * 2 * 64 * 8);
  for (int b = 0; b < ns; ++b)
    for (int st = 0; st < kF16StepsE; ++st)
      for (int n = 0; n < 2; ++n)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int kq = lane >> 4, tr = 4 * (st & 1) + (((kq & 1) << 1) | (kq >> 1)), tc = st >> 1, co = n * 16 + (lane & 15);
            const float wv = tr < 7 ? ldexpf(w->conv1_w[(((size_t)b * 32 + co) * 8 + j) * 49 + tr * 7 + tc], wex1[(size_t)b * 32 + co]) : 0.f;
            const _Float16 h = (_Float16)wv;
            const size_t base = ((((size_t)b * kF16StepsE + st) * 2 + n) * 2) * 64 * 8;
            f16e[base + (size_t)lane * 8 + j] = h;
            f16e[base + 64 * 8 + (size_t)lane * 8 + j] = (_Float16)(wv - (float)h);
          }
  std::vector<_Float16> g16((size_t)ns * 4 * kF16Steps * 4 * 2 * 64 * 8);
  for (int b = 0; b < ns; ++b)
    for (int ck = 0; ck < 4; ++ck)
      for (int st = 0; st < kF16Steps; ++st)
        for (int n = 0; n < 4; ++n)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int tap = 4 * st + (lane >> 4), co = n * 16 + (lane & 15), ci = ck * 8 + j;
              const float wv = tap < 49 ? ldexpf(w->conv2_w[(((size_t)b * 64 + co) * 32 + ci) * 49 + tap], wex2[(size_t)b * 64 + co]) : 0.f;
              const _Float16 h = (_Float16)wv;
              const size_t base = (((((size_t)b * 4 + ck) * kF16Steps + st) * 4 + n) * 2) * 64 * 8;
              g16[base + (size_t)lane * 8 + j] = h;
              g16[base + 64 * 8 + (size_t)lane * 8 + j] = (_Float16)(wv - (float)h);
            }
  std::vector<float> w1n((size_t)ns * 32);   // L1 norms of the conv1 filters, with a margin for the hi/lo rounding
  for (int b = 0; b < ns; ++b)
    for (int co = 0; co < 32; ++co) {
      double a = 0.0;
      for (int i = 0; i < 8 * 49; ++i) a += fabs((double)w->conv1_w[((size_t)b * 32 + co) * 392 + i]);
      w1n[(size_t)b * 32 + co] = (float)(a * 1.001);
    }
  int rc = 0;
  {
    _Float16 *d16 = nullptr, *e16 = nullptr, *d16e = nullptr;
    rc = mst::upload(&d16, f16.data(), f16.size());
    if (!rc) rc = mst::upload(&e16, g16.data(), g16.size());
    if (!rc) rc = mst::upload(&d16e, f16e.data(), f16e.size());
    e->w1frag16 = d16, e->w2frag16 = e16, e->w1frag16e = d16e;
  }
#define UP(dst, vec) if (!rc) rc = mst::upload(&e->dst, (vec).data(), (vec).size())
#define UPP(dst, ptr, n) if (!rc) rc = mst::upload(&e->dst, ptr, (size_t)(n))
  UP(w1norm, w1n); UP(f16_winv1, winv1); UP(f16_winv2, winv2);
  UP(w1frag, f1); UP(w2frag, f2); UP(s1, s1); UP(t1, t1); UP(s2, s2); UP(t2, t2);
  UP(w0t, w0t); UP(w3t, w3t); UP(hwt, hwt); UP(projfrag, pfrag); UP(att0frag, af);
  UPP(b0, w->mlp0_b, H); UPP(b3, w->mlp3_b, H); UPP(hb, w->head_b, ns * 192);
  UPP(att0_b, w->att0_b, A); UPP(att2_w, w->att2_w, A); UPP(proj_b, w->proj_b, E); UPP(att2_b, w->att2_b, 1);
  UPP(c1b, w->conv1_b, ns * 32); UPP(bn1w, w->bn1_w, ns * 32); UPP(bn1b, w->bn1_b, ns * 32);
  UPP(c2b, w->conv2_b, ns * 64); UPP(bn2w, w->bn2_w, ns * 64); UPP(bn2b, w->bn2_b, ns * 64);
  if (!rc) {   // input-gradient fragments of conv2, built on the device from a temporary copy of the weights
    float* tmp = nullptr;
    rc = mst::upload(&tmp, w->conv2_w, (size_t)ns * 64 * 32 * 49);
    if (!rc && hipMalloc(&e->w2dfrag, (size_t)ns * 16 * ConvGeom<3, 2>::WBP * sizeof(float)) != hipSuccess) rc = MST_ENOMEM;
    if (!rc) {
      (void)hipMemset(e->w2dfrag, 0, (size_t)ns * 16 * ConvGeom<3, 2>::WBP * sizeof(float));
      const long long t3 = (long long)ns * 16 * 49 * 2 * 64;
      hipLaunchKernelGGL(dgrad_fragments_kernel, dim3((unsigned)((t3 + 255) / 256)), dim3(256), 0, 0, tmp, e->w2dfrag, ns, ConvGeom<3, 2>::WBP);
      (void)hipDeviceSynchronize();
    }
    (void)hipFree(tmp);
  }
#undef UP
#undef UPP
  if (rc) {
    mst_encoder_destroy(e);
    return rc;
  }
  *out = e;
  return MST_OK;
}

void mst_encoder_destroy(mst_encoder* e) {
  if (!e) return;
  float* ptrs[] = {e->w1frag, e->w2frag, e->s1, e->t1, e->s2, e->t2, e->w0t, e->b0, e->w3t, e->b3, e->hwt,
                   e->hb, e->att0frag, e->att0_b, e->att2_w, e->projfrag, e->proj_b, e->c1b, e->bn1w, e->bn1b, e->c2b,
                   e->bn2w, e->bn2b, e->w2dfrag, e->att2_b, e->f16_wsc1e, e->f16_wsc2e, e->w2norm_e};
  for (float* q : ptrs) (void)hipFree(q);
  (void)hipFree(e->w1frag16), (void)hipFree(e->w1frag16e);
  (void)hipFree(e->w1norm), (void)hipFree(e->f16_winv1), (void)hipFree(e->f16_winv2);
  (void)hipFree(e->w2frag16);
  (void)hipFree(e->w2dfrag16);
  (void)hipFree(e->f16_wsc1), (void)hipFree(e->f16_wsc2), (void)hipFree(e->f16_wsc2d), (void)hipFree(e->f16_winv2d);
  (void)hipFree(e->w2norm);
  delete e;
}

int mst_encoder_set_precision(mst_encoder* e, int conv1_f16x3) {
  MST_REQUIRE(e, "mst_encoder_set_precision: NULL encoder");
  MST_REQUIRE(conv1_f16x3 == 0 || (conv1_f16x3 >= 1 && conv1_f16x3 <= 3 && (e->sub == 2 || (e->sub == 1 && e->cfg.split_size % 2 == 0))),
              "mst_encoder_set_precision: the f16 modes (1, 2, 3) need 2-row conv1 tiles (20-mel sub-bands, or an even split_size below 20)");
  e->conv1_f16x3 = conv1_f16x3;
  return MST_OK;
}

int mst_encoder_set_train_precision(mst_encoder* e, int f16_operands) {
  MST_REQUIRE(e, "mst_encoder_set_train_precision: NULL encoder");
  MST_REQUIRE(f16_operands == 0 || ((f16_operands == 1 || f16_operands == 2) && (e->sub == 2 || e->cfg.split_size % 2 == 0)),
              "mst_encoder_set_train_precision: modes 1 (f16 operands) and 2 (split precision) need 2-row conv1 tiles (20-mel sub-bands, or an even split_size below 20)");
  if (f16_operands && !e->w2dfrag16) {
    const int ns = e->cfg.n_subbands;
    bool ok = hipMalloc(&e->w2dfrag16, (size_t)ns * 8 * kF16Steps * 2 * 2 * 64 * 8 * sizeof(_Float16)) == hipSuccess;   // room for hi + lo
    ok = ok && hipMalloc(&e->f16_wsc1, (size_t)ns * 32 * 4) == hipSuccess && hipMalloc(&e->f16_wsc2, (size_t)ns * 64 * 4) == hipSuccess;
    ok = ok && hipMalloc(&e->f16_wsc2d, (size_t)ns * 32 * 4) == hipSuccess && hipMalloc(&e->f16_winv2d, (size_t)ns * 32 * 4) == hipSuccess;
    ok = ok && hipMalloc(&e->w2norm, (size_t)ns * 64 * 4) == hipSuccess;
    if (!ok) return mst::fail(MST_ENOMEM, "mst_encoder_set_train_precision: out of device memory");
  }
  e->train_f16 = f16_operands;
  return MST_OK;
}

size_t mst_encoder_workspace_bytes(const mst_encoder* e, int B, int frames) {
  if (!e || B <= 0 || frames < 20) return 0;
  return ws_layout(e, B, frames).total;
}

static bool conv1_resident_geometry(const mst_encoder* e) {
  return (e->sub == 2 || (e->sub == 1 && e->cfg.split_size % 2 == 0)) && !getenv("MST_CONV1_CHUNKED");
}

int mst_encoder_layout_supported(const mst_encoder* e, int layout) {
  if (!e) return 0;
  if (layout == MST_LOGMEL_REF) return 1;
  if (e->sub > 2) return 0;   // conv1_generic_kernel reads the reference layout
  if (layout == MST_LOGMEL_CM32) return e->conv1_f16x3 == 0 && conv1_resident_geometry(e);
  if (layout == MST_LOGMEL_CM16) return e->conv1_f16x3 != 0;
  return 0;
}

int mst_encoder_forward(const mst_encoder* e, const float* logmel, int frames, const float* feats, int B, float* emb,
                        const mst_encoder_taps* taps, void* workspace, size_t workspace_bytes, void* stream) {
  mst_logmel_in in{};
  in.layout = MST_LOGMEL_REF, in.data = logmel;
  return mst_encoder_forward_in(e, &in, frames, feats, B, emb, taps, workspace, workspace_bytes, stream);
}

int mst_encoder_forward_in(const mst_encoder* e, const mst_logmel_in* lin, int frames, const float* feats, int B, float* emb,
                           const mst_encoder_taps* taps, void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(e && lin && lin->data && feats && emb, "mst_encoder_forward: NULL argument");
  const int lay = lin->layout;
  MST_REQUIRE(mst_encoder_layout_supported(e, lay),
              "mst_encoder_forward: log-mel layout %d does not fit this encoder's conv1 kernel (precision mode %d; query "
              "mst_encoder_layout_supported)", lay, e->conv1_f16x3);
  MST_REQUIRE(lay != MST_LOGMEL_CM16 || e->conv1_f16x3 == 3 || lin->lo, "mst_encoder_forward: MST_LOGMEL_CM16 needs the low parts in the split-precision modes");
  MST_REQUIRE(lay != MST_LOGMEL_CM16 || e->conv1_f16x3 < 2 || lin->absmax, "mst_encoder_forward: MST_LOGMEL_CM16 needs absmax (stage A's per-clip max |log-mel|) when conv2 runs on float16 too");
  MST_REQUIRE(lay == MST_LOGMEL_REF || ((reinterpret_cast<uintptr_t>(lin->data) | reinterpret_cast<uintptr_t>(lin->lo)) & 15) == 0,
              "mst_encoder_forward: channel-minor log-mel must be 16-byte aligned");
  const float* logmel = static_cast<const float*>(lin->data);
  MST_REQUIRE(B > 0 && frames >= 20, "mst_encoder_forward: need B>0 and frames>=20 (B=%d frames=%d)", B, frames);
  const WsLayout L = ws_layout(e, B, frames);
  MST_REQUIRE(L.W2 >= 1, "mst_encoder_forward: clip too short for two pooling stages (frames=%d)", frames);
  if (!workspace || workspace_bytes < L.total)
    return mst::fail(MST_ENOMEM, "mst_encoder_forward: workspace %zu B < required %zu B", workspace_bytes, L.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int ns = e->cfg.n_subbands;
  float* film = (taps && taps->film) ? taps->film : reinterpret_cast<float*>(ws + L.film);
  float* pool1 = (taps && taps->pool1) ? taps->pool1 : reinterpret_cast<float*>(ws + L.pool1);
  float* pool_in = (taps && taps->pool_in) ? taps->pool_in : reinterpret_cast<float*>(ws + L.pool_in);
  float2* aff1 = reinterpret_cast<float2*>(ws + L.aff1);
  float2* aff2 = reinterpret_cast<float2*>(ws + L.aff2);
  float* scores = reinterpret_cast<float*>(ws + L.scores);
  auto mark = [&](int i) {
    if (taps && taps->events[i]) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(taps->events[i]), st);
  };
  mark(0);
  {
    FilmParams fp{feats, e->w0t, e->b0, e->w3t, e->b3, e->hwt, e->hb, e->s1, e->t1, e->s2, e->t2,
                  film, aff1, aff2, e->cfg.feature_dim, e->cfg.film_hidden, ns};
    const int groups = ns < 4 ? ns : 4;
    const int bpg = (ns + groups - 1) / groups;
    const size_t lds = (size_t)(e->cfg.feature_dim + 2 * e->cfg.film_hidden + bpg * 192) * sizeof(float);
    hipLaunchKernelGGL(film_kernel, dim3(B, groups), dim3(256), lds, st, fp);
    MST_HIP_CHECK(hipGetLastError());
  }
  mark(1);
  const int grid = e->num_cus;
  {
    ConvParams cp{};
    cp.in = logmel, cp.wfrag = e->w1frag, cp.aff = aff1, cp.out = pool1, cp.B = B, cp.nsub = ns;
    cp.in_rows = e->cfg.split_size, cp.in_cols = frames;
    cp.in_cstride = e->cfg.n_mels * frames;
    cp.in_bandoff = e->cfg.overlap * frames;
    cp.in_clipstride = (long long)8 * e->cfg.n_mels * frames;
    cp.in_lo = lin->lo, cp.cm_mels = e->cfg.n_mels, cp.cm_overlap = e->cfg.overlap;
    cp.out_rows = e->H1, cp.out_cols = L.W1;
    cp.tiles_r = e->H1;
    cp.tiles_c = e->sub == 2 ? (L.W1 + 7) / 8 : (L.W1 + 15) / 16;
    cp.sets_per_band = (B * cp.tiles_r * cp.tiles_c + kConvWaves - 1) / kConvWaves;
    const int g = std::min(grid, ns * cp.sets_per_band);
    hipError_t err;
    if (e->sub > 2) {   // first-pool heights >= 3: the generic kernel (reference layout, exact fp32)
      MST_REQUIRE(lay == MST_LOGMEL_REF && e->conv1_f16x3 == 0, "mst_encoder_forward: split_size=%d runs in the reference layout, fp32 only", e->cfg.split_size);
      const size_t lds = (size_t)(392 * 32 + 8 * (e->cfg.split_size + 6) * kGenPC) * sizeof(float);
      static unsigned long long attr_gen = 0;   // per-device bit mask: the attribute belongs to the device
      if (mst::first_use_on_device(attr_gen)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (err != hipSuccess) return mst::fail(MST_EHIP, "conv1 generic attribute failed: %s", hipGetErrorString(err));
      }
      hipLaunchKernelGGL(conv1_generic_kernel, dim3((L.W1 + kGenCols - 1) / kGenCols, B * ns), dim3(256), lds, st, cp, e->sub);
      err = hipGetLastError();
    } else if (e->conv1_f16x3) {   // (set_precision admits the f16 modes only for geometries that fit the 2 x 40 tiles)
      using C = CC<1, 2>;
      if (e->sub == 1) {   // 16-mel sub-bands: pool height 1 on the same tiles
        cp.pool_h = 1;
        cp.tiles_r = e->cfg.split_size / 2;
        cp.tiles_c = (L.W1 + 7) / 8;
        cp.sets_per_band = (B * cp.tiles_r * cp.tiles_c + kConvWaves - 1) / kConvWaves;
      }
      const int g = std::min(grid, ns * cp.sets_per_band);
      constexpr size_t lds = (size_t)(kF16Steps * C::NT * 2 * 64 + kConvWaves * 2 * C::PR * C::PC) * 16;
      constexpr size_t lds_e3 = (size_t)(kF16StepsE * C::NT * 2 * 64 + kConvWaves * 2 * kF16ePatch) * 16;   // conv1_f16e_kernel
      constexpr size_t lds_e1 = (size_t)(kF16StepsE * C::NT * 64 + kF16eWaves<1> * kF16ePatch) * 16;
      static_assert(lds_e3 <= 160 * 1024, "conv1_f16e_kernel<3> must fit one CU's LDS");
      static unsigned long long attr16 = 0;   // per-device bit mask: the attribute belongs to the device
      if (mst::first_use_on_device(attr16)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x3_kernel<2, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
          err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x3_kernel<2, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
          err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16e_kernel<3>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e3);
        if (err == hipSuccess)
          err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16e_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e1);
        if (err != hipSuccess) return mst::fail(MST_EHIP, "conv1 f16x3 attribute failed: %s", hipGetErrorString(err));
      }
      const bool both = e->conv1_f16x3 >= 2;
      cp.f16_winv = e->f16_winv1;
      if (both && !(taps && taps->pool1)) cp.out = nullptr;   // fp32 pool1 only when a tap asks for it
      if (both) {   // range scale of conv2's f16 input from a rigorous bound (see f16_scale_kernel): no host check, no refusal
        const unsigned* xmax = lin->absmax;   // stage A's per-clip max |log-mel|, when the caller has it
        float* fsc = reinterpret_cast<float*>(ws + L.f16scale);
        if (!xmax) {   // (reference-layout input only, checked above): one more pass over the log-mel
          unsigned* xm = reinterpret_cast<unsigned*>(ws + L.xmax);
          MST_HIP_CHECK(hipMemsetAsync(xm, 0, (size_t)B * sizeof(unsigned), st));
          const long long npc = (long long)8 * e->cfg.n_mels * frames;
          hipLaunchKernelGGL(absmax_kernel, dim3(64, B), dim3(256), 0, st, logmel, npc, xm);
          xmax = xm;
        }
        hipLaunchKernelGGL(f16_scale_kernel, dim3(B * ns), dim3(64), 0, st, aff1, e->w1norm, xmax, fsc, ns);
        cp.f16_scale = fsc;
      }
      const h16x8* wf = reinterpret_cast<const h16x8*>(e->w1frag16);
      _Float16* oh = both ? reinterpret_cast<_Float16*>(ws + L.pool1_h16) : nullptr;
      _Float16* ol = e->conv1_f16x3 == 2 ? reinterpret_cast<_Float16*>(ws + L.pool1_l16) : nullptr;
      if (lay == MST_LOGMEL_CM16) {   // the bank-conflict-free build (its own k order: w1frag16e)
        const h16x8* wfe = reinterpret_cast<const h16x8*>(e->w1frag16e);
        if (e->conv1_f16x3 == 3) {   // plain f16: 12 waves per workgroup, sets of 12 tiles
          cp.sets_per_band = (B * cp.tiles_r * cp.tiles_c + kF16eWaves<1> - 1) / kF16eWaves<1>;
          const int g1 = std::min(grid, ns * cp.sets_per_band);
          hipLaunchKernelGGL((conv1_f16e_kernel<1>), dim3(g1), dim3(kF16eWaves<1> * 64), lds_e1, st, cp, wfe, oh, ol);
        }
        else hipLaunchKernelGGL((conv1_f16e_kernel<3>), dim3(g), dim3(kConvThreads), lds_e3, st, cp, wfe, oh, ol);
      } else if (e->conv1_f16x3 == 3) hipLaunchKernelGGL((conv1_f16x3_kernel<2, 1>), dim3(g), dim3(kConvThreads), lds, st, cp, wf, oh, ol);
      else hipLaunchKernelGGL((conv1_f16x3_kernel<2, 3>), dim3(g), dim3(kConvThreads), lds, st, cp, wf, oh, ol);
      err = hipGetLastError();
    } else if (conv1_resident_geometry(e)) {
      // band-resident kernel on 2 x 40 tiles; 16-mel sub-bands (pool height 1) take it with two 1 x 5 windows per lane
      using C = CC<1, 2>;
      if (e->sub == 1) {
        cp.pool_h = 1;
        cp.tiles_r = e->cfg.split_size / 2;
        cp.tiles_c = (L.W1 + 7) / 8;
        cp.sets_per_band = (B * cp.tiles_r * cp.tiles_c + kConvWaves - 1) / kConvWaves;
      }
      if (!getenv("MST_CONV1_BAND_MAJOR")) {   // eval forward: the bands of a clip group next to each other (decode() of the kernel)
        const char* env = getenv("MST_CONV1_CLIP_GROUP");
        cp.clip_major = std::max(1, std::min(B, env ? atoi(env) : 1));   // 1: every (clip, band) patch comes from HBM exactly once (measured)
        cp.sets_per_band = (cp.clip_major * cp.tiles_r * cp.tiles_c + kConvWaves - 1) / kConvWaves;
      }
      const int g = std::min(grid, (cp.clip_major ? (B + cp.clip_major - 1) / cp.clip_major : 1) * ns * cp.sets_per_band);
      constexpr size_t lds = (size_t)(2 * 49 * C::NT * 64 + kConvWaves * 8 * C::PR * C::PC) * sizeof(float);
      static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
      if (mst::first_use_on_device(attr_set)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_resident_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
          err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_resident_kernel<2, 0, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return mst::fail(MST_EHIP, "conv1 attribute failed: %s", hipGetErrorString(err));
      }
#ifdef MST_TRACE
      static float* trace_buf = nullptr;
      if (!trace_buf) (void)hipMalloc(&trace_buf, 256 * kConvWaves * 8 * sizeof(float));
      cp.yraw = trace_buf;
#endif
      if (lay == MST_LOGMEL_CM32) hipLaunchKernelGGL((conv1_resident_kernel<2, 0, 1>), dim3(g), dim3(kConvThreads), lds, st, cp);
      else hipLaunchKernelGGL((conv1_resident_kernel<2>), dim3(g), dim3(kConvThreads), lds, st, cp);
#ifdef MST_TRACE
      {
        static int calls = 0;
        if (++calls == 10) {
          (void)hipDeviceSynchronize();
          std::vector<float> h(256 * kConvWaves * 8);
          (void)hipMemcpy(h.data(), trace_buf, h.size() * sizeof(float), hipMemcpyDeviceToHost);
          for (int w = 0; w < kConvWaves; ++w) {   // average over the workgroups, per wave index
            double a[6] = {0, 0, 0, 0, 0, 0};
            for (int b = 0; b < g; ++b)
              for (int i = 0; i < 6; ++i) a[i] += h[(b * kConvWaves + w) * 8 + i] / g;
            fprintf(stderr, "trace wave %d: tiles %.1f  k-steps %.0f (%.0f per tile)  wait for the patch loads %.0f (%.0f)  stage %.0f (%.0f)  barriers %.0f  total %.0f\n",
                    w, a[4], a[0], a[0] / a[4], a[1], a[1] / a[4], a[2], a[2] / a[4], a[3], a[5]);
          }
        }
      }
#endif
      err = hipGetLastError();
    } else {
      err = e->sub == 2 ? launch_conv<1, 2>(cp, g, st) : launch_conv<1, 1>(cp, g, st);
    }
    if (err != hipSuccess) return mst::fail(MST_EHIP, "conv1 launch failed: %s", hipGetErrorString(err));
  }
  mark(2);
  {
    ConvParams cp{};
    cp.in = pool1, cp.wfrag = e->w2frag, cp.aff = aff2, cp.out = pool_in, cp.B = B, cp.nsub = ns;
    cp.in_rows = e->H1, cp.in_cols = L.W1;
    cp.in_cstride = e->H1 * L.W1;
    cp.in_bandoff = 32 * e->H1 * L.W1;
    cp.in_clipstride = (long long)ns * 32 * e->H1 * L.W1;
    cp.out_rows = e->FD, cp.out_cols = L.W2;
    cp.tiles_r = (e->FD + 1) / 2;
    cp.tiles_c = (L.W2 + 1) / 2;
    cp.sets_per_band = (B * cp.tiles_r * cp.tiles_c + kConvWaves - 1) / kConvWaves;
    const int g = std::min(grid, ns * cp.sets_per_band);
    hipError_t err;
    if (e->conv1_f16x3 >= 2) {
      constexpr size_t lds = (size_t)(kF16Steps * 4 * 2 * 64 + kConvWaves * 2 * 14 * 14) * 16;
      static unsigned long long attr = 0;   // per-device bit mask: the attribute belongs to the device
      if (mst::first_use_on_device(attr)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_f16x3_kernel<3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
          err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_f16x3_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return mst::fail(MST_EHIP, "conv2 f16x3 attribute failed: %s", hipGetErrorString(err));
      }
      cp.f16_scale = reinterpret_cast<const float*>(ws + L.f16scale);
      cp.f16_winv = e->f16_winv2;
      const h16x8* ih = reinterpret_cast<const h16x8*>(ws + L.pool1_h16);
      const h16x8* il = reinterpret_cast<const h16x8*>(ws + L.pool1_l16);
      const h16x8* wf = reinterpret_cast<const h16x8*>(e->w2frag16);
      if (e->conv1_f16x3 == 3) hipLaunchKernelGGL((conv2_f16x3_kernel<1>), dim3(g), dim3(kConvThreads), lds, st, cp, ih, il, wf);
      else hipLaunchKernelGGL((conv2_f16x3_kernel<3>), dim3(g), dim3(kConvThreads), lds, st, cp, ih, il, wf);
      err = hipGetLastError();
    } else {
#ifdef MST_TRACE
      static float* trace_buf2 = nullptr;
      if (!trace_buf2) (void)hipMalloc(&trace_buf2, 256 * kConvWaves * 8 * sizeof(float));
      cp.yraw = trace_buf2;
#endif
      err = launch_conv<2, 2>(cp, g, st);
#ifdef MST_TRACE
      {
        static int calls = 0;
        if (++calls == 10) {
          (void)hipDeviceSynchronize();
          std::vector<float> h(256 * kConvWaves * 8);
          (void)hipMemcpy(h.data(), trace_buf2, h.size() * sizeof(float), hipMemcpyDeviceToHost);
          for (int w = 0; w < kConvWaves; ++w) {
            double a[6] = {0, 0, 0, 0, 0, 0};
            for (int b = 0; b < g; ++b)
              for (int i = 0; i < 6; ++i) a[i] += h[(b * kConvWaves + w) * 8 + i] / g;
            fprintf(stderr, "conv2 trace wave %d: chunks %.1f  staging %.0f (%.0f per chunk)  barrier wait %.0f (%.0f)  taps %.0f (%.0f)  total %.0f\n",
                    w, a[4], a[0], a[0] / a[4], a[1], a[1] / a[4], a[2], a[2] / a[4], a[5]);
          }
        }
      }
#endif
    }
    if (err != hipSuccess) return mst::fail(MST_EHIP, "conv2 launch failed: %s", hipGetErrorString(err));
  }
  mark(3);
  {
    AttnParams ap{pool_in, e->att0frag, e->att0_b, e->att2_w, e->att2_b, scores, B, e->C, L.W2, e->cfg.attn_hidden};
    const int mtiles = (B * L.W2 + 16 * kAttnMT - 1) / (16 * kAttnMT);
    hipLaunchKernelGGL(attn_scores_kernel, dim3(mtiles), dim3(256), 0, st, ap);
    MST_HIP_CHECK(hipGetLastError());
  }
  mark(4);
  {
    float* pooled = reinterpret_cast<float*>(ws + L.pooled);
    PoolParams pp{pool_in, scores, pooled, e->C, L.W2};
    const int slices = (e->C + 127) / 128;
    hipLaunchKernelGGL(attn_pool_kernel, dim3(B, slices), dim3(256), (size_t)((L.W2 + 3) & ~3) * sizeof(float), st, pp);
    MST_HIP_CHECK(hipGetLastError());
    ProjParams pj{pooled, e->projfrag, e->proj_b, emb, B, e->C, e->cfg.embed_dim};
    const int mt_all = (B + 15) / 16;   // row tiles of clips
    if (mt_all <= 1) launch_proj<1>(pj, 1, st);
    else if (mt_all <= 2) launch_proj<2>(pj, 1, st);
    else if (mt_all <= 3) launch_proj<3>(pj, 1, st);
    else if (mt_all <= 5) launch_proj<5>(pj, 1, st);
    else launch_proj<8>(pj, (B + 127) / 128, st);
    MST_HIP_CHECK(hipGetLastError());
  }
  mark(5);
  return MST_OK;
}

// ---- training forward ---------------------------------------------------------------------------------------------
namespace {
struct TrainLayout {
  WsLayout base;
  size_t y1, y2, stats1, stats2, bn1, bn2, dfilm_acc, dw_acc, total;
  size_t t_pool1_h16, t_pool1_l16, t_f16scale, t_xmax, t_bscale, t_dyg1, t_dyg2, t_ys1, t_ys2;   // f16 training modes only (0 bytes otherwise)
  int tr1, tc1, tr2, tc2;
};
TrainLayout train_layout(const mst_encoder* e, int B, int frames) {
  TrainLayout T{};
  T.base = ws_layout(e, B, frames);
  const int ns = e->cfg.n_subbands;
  const bool two_row = e->sub == 2 || train_fwd16(e);          // the f16 kernels tile 16-mel sub-bands like 20-mel ones: 2 x 40
  const int tcols1 = two_row ? 40 : 80;                        // conv1 tiles: 2 x 40 or 1 x 80, ALL columns (statistics)
  T.tr1 = two_row ? e->cfg.split_size / 2 : e->H1, T.tc1 = (frames + tcols1 - 1) / tcols1;
  T.tr2 = (e->H1 + 7) / 8, T.tc2 = (T.base.W1 + 7) / 8;      // conv2 tiles: 8 x 8, ALL rows
  size_t o = T.base.total;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += mst::align_up(bytes, 256);
    return at;
  };
  const size_t ybytes = e->train_f16 == 1 ? 2 : 4;   // mode 1 stores the raw conv outputs as float16
  T.y1 = take((size_t)B * ns * T.tr1 * T.tc1 * 2 * 64 * 20 * ybytes);
  T.y2 = take((size_t)B * ns * T.tr2 * T.tc2 * 4 * 64 * 16 * ybytes);
  T.stats1 = take((size_t)(ns * 32 * 2 + 1) * sizeof(mst::DetAcc));   // + the clip-count word that rides with the sums
  T.stats2 = take((size_t)(ns * 64 * 2 + 1) * sizeof(mst::DetAcc));
  T.bn1 = take((size_t)ns * 32 * 8);
  T.bn2 = take((size_t)ns * 64 * 8);
  T.dfilm_acc = take((size_t)B * ns * 192 * sizeof(mst::DetAcc));
  T.dw_acc = take((size_t)ns * 64 * 1568 * sizeof(mst::DetAcc));   // weight-gradient accumulators (conv2's size; conv1 reuses it)
  const size_t f = train_fwd16(e) ? 1 : 0, fb = train_bwd16(e) ? 1 : 0;
  T.t_pool1_h16 = take(f * (size_t)B * ns * 32 * e->H1 * T.base.W1 * 2);   // pool1 as f16, channel-minor (conv2's operand)
  T.t_pool1_l16 = take((e->train_f16 == 2 ? 1 : 0) * (size_t)B * ns * 32 * e->H1 * T.base.W1 * 2);   // its low part (split precision)
  T.t_f16scale = take(f * (size_t)B * ns * 2 * 4);
  T.t_xmax = take(f * (size_t)B * 4);
  T.t_ys1 = take((e->train_f16 == 1 ? 1 : 0) * (size_t)ns * 2 * 4);              // range scales of the stored f16 conv outputs, [nsub][2]
  T.t_ys2 = take((e->train_f16 == 1 ? 1 : 0) * (size_t)ns * 2 * 4);
  T.t_bscale = take(fb * 16);                                               // (s, 1/s) of the backward pass + the max |d pool_in| bits
  const size_t cgs = (size_t)((B + 7) / 8);                                 // d(conv output) as f16 in the weight gradients' operand layout
  const size_t hl = e->train_f16 == 2 ? 2 : 1;                              // split precision: hi and lo
  T.t_dyg1 = take(fb * hl * (size_t)ns * cgs * T.tr1 * T.tc1 * 2 * 20 * 64 * 16);
  T.t_dyg2 = take(fb * hl * (size_t)ns * cgs * T.tr2 * T.tc2 * 4 * 16 * 64 * 16);
  T.total = o;
  return T;
}
}  // namespace

size_t mst_encoder_train_workspace_bytes(const mst_encoder* e, int B, int frames) {
  if (!e || B <= 0 || frames < 20 || e->sub > 2) return 0;   // (the training kernels cover first-pool heights 1 and 2)
  return train_layout(e, B, frames).total;
}

int mst_encoder_train_stats_buffer(const mst_encoder* e, int layer, int B, int frames, size_t* offset_bytes, size_t* n_int64) {
  MST_REQUIRE(e && offset_bytes && n_int64 && (layer == 1 || layer == 2) && B > 0 && frames >= 20,
              "mst_encoder_train_stats_buffer: bad arguments");
  const TrainLayout T = train_layout(e, B, frames);
  *offset_bytes = layer == 1 ? T.stats1 : T.stats2;
  // [n_sub][C][2 sums][2 words] + one more pair whose first word is this rank's clip count: summed with the rest, it gives
  // every rank the global count, so ranks with different numbers of clips (a ragged last batch) normalise identically
  *n_int64 = ((size_t)e->cfg.n_subbands * (layer == 1 ? 32 : 64) * 2 + 1) * (sizeof(mst::DetAcc) / 8);
  return MST_OK;
}

int mst_encoder_train_scale_buffer(const mst_encoder* e, int B, int frames, size_t* offset_bytes) {
  MST_REQUIRE(e && offset_bytes && B > 0 && frames >= 20, "mst_encoder_train_scale_buffer: bad arguments");
  MST_REQUIRE(train_bwd16(e), "mst_encoder_train_scale_buffer: the fp32 training mode has no internal loss scale");
  *offset_bytes = train_layout(e, B, frames).t_bscale + 2 * sizeof(float);
  return MST_OK;
}

int mst_encoder_train_layout_supported(const mst_encoder* e, int layout) {
  if (!e) return 0;
  if (layout == MST_LOGMEL_REF) return 1;
  // the float16 training kernels (forward conv1 and its weight gradient) read stage A's float16 planes directly; the fp32
  // training kernels keep the reference layout
  if (layout == MST_LOGMEL_CM16) return train_fwd16(e) ? 1 : 0;
  return 0;
}

int mst_encoder_forward_train(const mst_encoder* e, const float* logmel, int frames, const float* feats, int B, float* emb,
                              const mst_encoder_train_taps* taps, void* workspace, size_t workspace_bytes, void* stream) {
  mst_logmel_in in{};
  in.layout = MST_LOGMEL_REF, in.data = logmel;
  return mst_encoder_forward_train_in(e, &in, frames, feats, B, emb, taps, workspace, workspace_bytes, stream);
}

int mst_encoder_forward_train_in(const mst_encoder* e, const mst_logmel_in* lin, int frames, const float* feats, int B, float* emb,
                                 const mst_encoder_train_taps* taps, void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(e && e->sub <= 2, "mst_encoder_forward_train: the training kernels cover first-pool heights 1 and 2 (split_size < 30)");
  MST_REQUIRE(e && lin && lin->data && (feats || (taps && taps->film_in)), "mst_encoder_forward_train: NULL argument");
  const int lay = lin->layout;
  MST_REQUIRE(mst_encoder_train_layout_supported(e, lay),
              "mst_encoder_forward_train: log-mel layout %d does not fit the training kernels of precision mode %d (query "
              "mst_encoder_train_layout_supported)", lay, e->train_f16);
  MST_REQUIRE(lay != MST_LOGMEL_CM16 || e->train_f16 == 1 || lin->lo, "mst_encoder_forward_train: MST_LOGMEL_CM16 needs the low parts in the split-precision mode");
  MST_REQUIRE(lay == MST_LOGMEL_REF || ((reinterpret_cast<uintptr_t>(lin->data) | reinterpret_cast<uintptr_t>(lin->lo)) & 15) == 0,
              "mst_encoder_forward_train: channel-minor log-mel must be 16-byte aligned");
  const float* logmel = static_cast<const float*>(lin->data);
  MST_REQUIRE(B > 0 && frames >= 20, "mst_encoder_forward_train: need B>0 and frames>=20 (B=%d frames=%d)", B, frames);
  const TrainLayout T = train_layout(e, B, frames);
  const WsLayout& L = T.base;
  MST_REQUIRE(L.W2 >= 1, "mst_encoder_forward_train: clip too short for two pooling stages (frames=%d)", frames);
  if (!workspace || workspace_bytes < T.total)
    return mst::fail(MST_ENOMEM, "mst_encoder_forward_train: workspace %zu B < required %zu B", workspace_bytes, T.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int ns = e->cfg.n_subbands;
  float* film = reinterpret_cast<float*>(ws + L.film);   // kept in the workspace: the backward pass reads it
  float* pool1 = (taps && taps->pool1) ? taps->pool1 : reinterpret_cast<float*>(ws + L.pool1);
  float* pool_in = (taps && taps->pool_in) ? taps->pool_in : reinterpret_cast<float*>(ws + L.pool_in);
  float2* aff1 = reinterpret_cast<float2*>(ws + L.aff1);
  float2* aff2 = reinterpret_cast<float2*>(ws + L.aff2);
  float* scores = reinterpret_cast<float*>(ws + L.scores);
  float* y1 = reinterpret_cast<float*>(ws + T.y1);
  float* y2 = reinterpret_cast<float*>(ws + T.y2);
  mst::DetAcc* stats1 = reinterpret_cast<mst::DetAcc*>(ws + T.stats1);
  mst::DetAcc* stats2 = reinterpret_cast<mst::DetAcc*>(ws + T.stats2);
  // phases (taps->phase; 0 = everything): 1 = FiLM + conv1 raw output + its statistics; 2 = BatchNorm 1 + FiLM + pooling + conv2
  // raw output + its statistics; 3 = BatchNorm 2 + FiLM + pooling + head.  Between the phases a data-parallel caller sums the
  // statistics accumulators over its ranks (mst_encoder_train_stats_buffer) and passes count_scale = world size.
  const int phase = taps ? taps->phase : 0;
  MST_REQUIRE(phase >= 0 && phase <= 3, "mst_encoder_forward_train: phase %d (0..3)", phase);
  const bool run_a = phase == 0 || phase == 1, run_b = phase == 0 || phase == 2, run_c = phase == 0 || phase == 3;
  const double cscale = (taps && taps->count_scale > 0.0) ? taps->count_scale : 1.0;
  const int grid = e->num_cus;
  hipError_t err;
  if (run_a) {
  // (stats1 and stats2 are neighbours in the workspace: one fill)
  MST_HIP_CHECK(hipMemsetAsync(stats1, 0, (T.stats2 - T.stats1) + (size_t)ns * 64 * 2 * sizeof(mst::DetAcc), st));
  if (phase != 0)   // cross-rank statistics: this rank's clip count rides with each layer's sums
    hipLaunchKernelGGL(set_clips_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<long long*>(stats1 + ns * 32 * 2),
                       reinterpret_cast<long long*>(stats2 + ns * 64 * 2), (long long)B);
  if (taps && taps->film_in) {   // FiLM parameters computed by the caller (its MLP keeps its autograd graph)
    MST_HIP_CHECK(hipMemcpyAsync(film, taps->film_in, (size_t)B * ns * 192 * 4, hipMemcpyDeviceToDevice, st));
  } else {   // FiLM MLP (its eval-mode affines are overwritten by bn_fold_kernel below)
    FilmParams fp{feats, e->w0t, e->b0, e->w3t, e->b3, e->hwt, e->hb, e->s1, e->t1, e->s2, e->t2,
                  film, aff1, aff2, e->cfg.feature_dim, e->cfg.film_hidden, ns};
    const int groups = ns < 4 ? ns : 4;
    const int bpg = (ns + groups - 1) / groups;
    const size_t lds = (size_t)(e->cfg.feature_dim + 2 * e->cfg.film_hidden + bpg * 192) * sizeof(float);
    hipLaunchKernelGGL(film_kernel, dim3(B, groups), dim3(256), lds, st, fp);
    MST_HIP_CHECK(hipGetLastError());
  }
  {   // conv1 raw + statistics
    ConvParams cp{};
    cp.in = logmel, cp.wfrag = e->w1frag, cp.B = B, cp.nsub = ns;
    cp.in_rows = e->cfg.split_size, cp.in_cols = frames;
    cp.in_cstride = e->cfg.n_mels * frames;
    cp.in_bandoff = e->cfg.overlap * frames;
    cp.in_clipstride = (long long)8 * e->cfg.n_mels * frames;
    cp.out_rows = e->H1, cp.out_cols = L.W1;
    cp.tiles_r = T.tr1, cp.tiles_c = T.tc1;
    cp.sets_per_band = (B * cp.tiles_r * cp.tiles_c + kConvWaves - 1) / kConvWaves;
    cp.yraw = y1, cp.stats = stats1, cp.bias = e->c1b, cp.raw_rows = e->cfg.split_size, cp.raw_cols = frames;
    cp.acc_tr = T.tr1, cp.acc_tc = T.tc1;
    const int g = std::min(grid, ns * cp.sets_per_band);
    if (train_fwd16(e)) {   // f16 operands (mode 1) or 3-term split precision (mode 2), fp32 accumulate (encoder_f16train.inc)
      using C = CC<1, 2>;
      unsigned* xmax = reinterpret_cast<unsigned*>(ws + T.t_xmax);   // max |log-mel| per clip: bounds for the range scales
      if (lin->absmax) {   // stage A found it while it wrote the log-mel
        MST_HIP_CHECK(hipMemcpyAsync(xmax, lin->absmax, (size_t)B * sizeof(unsigned), hipMemcpyDeviceToDevice, st));
      } else {
        MST_REQUIRE(lay == MST_LOGMEL_REF, "mst_encoder_forward_train: MST_LOGMEL_CM16 needs absmax (stage A's per-clip max |log-mel|)");
        MST_HIP_CHECK(hipMemsetAsync(xmax, 0, (size_t)B * sizeof(unsigned), st));
        hipLaunchKernelGGL(absmax_kernel, dim3(64, B), dim3(256), 0, st, logmel, (long long)8 * e->cfg.n_mels * frames, xmax);
      }
      cp.in_lo = lin->lo, cp.cm_mels = e->cfg.n_mels, cp.cm_overlap = e->cfg.overlap;
      if (e->train_f16 == 1) {   // the raw output is stored as float16 times a per-band power of two
        float* ys1 = reinterpret_cast<float*>(ws + T.t_ys1);
        hipLaunchKernelGGL(f16_yscale_kernel, dim3(ns), dim3(64), 0, st, e->w1norm, e->c1b, 32, xmax, B, static_cast<const float*>(nullptr), ys1);
        cp.yraw16 = reinterpret_cast<_Float16*>(y1), cp.y_scale = ys1;
      }
      constexpr size_t lds = (size_t)(kF16Steps * C::NT * 2 * 64 + kConvWaves * 2 * C::PR * C::PC) * 16;
      static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
      if (mst::first_use_on_device(attr_set)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x3_kernel<2, 1, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
          err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x3_kernel<2, 3, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
          err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x3_kernel<2, 1, 2, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
          err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x3_kernel<2, 3, 1, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return mst::fail(MST_EHIP, "conv1 (f16 train) attribute failed: %s", hipGetErrorString(err));
      }
      cp.f16_winv = e->f16_winv1;
      const h16x8* wf = reinterpret_cast<const h16x8*>(e->w1frag16);
      _Float16* none = nullptr;
      if (lay == MST_LOGMEL_CM16) {   // ready-made float16 operands: 16-byte copies instead of the fp32 -> f16 staging
        if (e->train_f16 == 2) hipLaunchKernelGGL((conv1_f16x3_kernel<2, 3, 1, 2>), dim3(g), dim3(kConvThreads), lds, st, cp, wf, none, none);
        else hipLaunchKernelGGL((conv1_f16x3_kernel<2, 1, 2, 2>), dim3(g), dim3(kConvThreads), lds, st, cp, wf, none, none);
      } else if (e->train_f16 == 2) hipLaunchKernelGGL((conv1_f16x3_kernel<2, 3, 1>), dim3(g), dim3(kConvThreads), lds, st, cp, wf, none, none);
      else hipLaunchKernelGGL((conv1_f16x3_kernel<2, 1, 2>), dim3(g), dim3(kConvThreads), lds, st, cp, wf, none, none);
    } else if (e->sub == 2) {
      using C = CC<1, 2>;
      constexpr size_t lds = (size_t)(2 * 49 * C::NT * 64 + kConvWaves * 8 * C::PR * C::PC) * sizeof(float);
      static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
      if (mst::first_use_on_device(attr_set)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_resident_kernel<2, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return mst::fail(MST_EHIP, "conv1 (train) attribute failed: %s", hipGetErrorString(err));
      }
      hipLaunchKernelGGL((conv1_resident_kernel<2, 1>), dim3(g), dim3(kConvThreads), lds, st, cp);
    } else {   // 10..19-mel sub-bands (pool height 1): the chunked kernel with the raw epilogue
      using GEO = ConvGeom<1, 1>;
      const size_t lds = (size_t)(2 * GEO::WBP + kConvWaves * GEO::PATCH) * sizeof(float);
      static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
      if (mst::first_use_on_device(attr_set)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel<1, 1, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return mst::fail(MST_EHIP, "conv1 (train) attribute failed: %s", hipGetErrorString(err));
      }
      hipLaunchKernelGGL((conv_kernel<1, 1, 1>), dim3(g), dim3(kConvThreads), lds, st, cp);
    }
    MST_HIP_CHECK(hipGetLastError());
  }
  }   // run_a
  if (run_b) {
  {
    float2* bnstat = reinterpret_cast<float2*>(ws + T.bn1);
    FoldParams fp{stats1, e->bn1w, e->bn1b, film, aff1, bnstat, cscale * B * e->cfg.split_size * frames, e->cfg.bn_eps,
                  ns, 32, 0, 32, phase ? reinterpret_cast<const long long*>(stats1 + ns * 32 * 2) : nullptr,
                  (double)e->cfg.split_size * frames};
    hipLaunchKernelGGL(bn_fold_kernel, dim3(ns, B), dim3(32), 0, st, fp);
    ApplyParams ap{y1, e->cfg.split_size, aff1, pool1, taps ? taps->drop1_mask : nullptr, taps ? taps->drop1_scale : 1.f,
                   B, ns, T.tr1, T.tc1, e->H1, L.W1, (long long)B * ns * T.tr1 * T.tc1 * 2 * 64, nullptr, nullptr, frames};
    if (train_fwd16(e)) {   // range scale of conv2's f16 operand: one power of two per band from a bound on the pooled values
      if (!(taps && taps->pool1)) ap.out = nullptr;   // conv2 and its weight gradient read the float16 planes: fp32 pool1 only on request
      unsigned* xmax = reinterpret_cast<unsigned*>(ws + T.t_xmax);   // (computed in front of conv1)
      float* fsc = reinterpret_cast<float*>(ws + T.t_f16scale);
      hipLaunchKernelGGL(f16_scale_band_kernel, dim3(ns), dim3(64), 0, st, aff1, e->w1norm, xmax, e->c1b,
                         (taps && taps->drop1_mask_out && taps->drop1_p > 0.f) ? 1.f / (1.f - taps->drop1_p)
                         : ((taps && taps->drop1_mask) ? taps->drop1_scale : 1.f), fsc, B, ns);
      ap.out_h16 = reinterpret_cast<_Float16*>(ws + T.t_pool1_h16), ap.f16_scale = fsc;
      if (e->train_f16 == 2) ap.out_l16 = reinterpret_cast<_Float16*>(ws + T.t_pool1_l16);
      if (e->train_f16 == 1) ap.y_scale = reinterpret_cast<const float*>(ws + T.t_ys1);
    }
    ap.pool_h = e->sub;   // 2 x 40 tiles with 16-mel sub-bands (f16 modes): MaxPool2d((1, 5))
    if (taps && taps->drop1_mask_out && taps->drop1_p > 0.f) {   // Dropout drawn in the kernel
      MST_REQUIRE(taps->drop1_p < 1.f, "mst_encoder_forward_train: drop1_p must be in [0, 1)");
      ap.mask_out = taps->drop1_mask_out, ap.seed = taps->drop1_seed;
      ap.drop_thresh = (unsigned)std::min(4294967295.0, (double)taps->drop1_p * 4294967296.0);
      ap.mask = nullptr, ap.mask_scale = 1.f / (1.f - taps->drop1_p);
    }
    if (e->sub == 2 || train_fwd16(e)) hipLaunchKernelGGL((apply_kernel<1, 2>), dim3((unsigned)((ap.units + 255) / 256)), dim3(256), 0, st, ap);
    else hipLaunchKernelGGL((apply_kernel<1, 1>), dim3((unsigned)((ap.units + 255) / 256)), dim3(256), 0, st, ap);
    MST_HIP_CHECK(hipGetLastError());
  }
  {   // conv2 raw + statistics (all 10 rows: rows 8, 9 never reach MaxPool(4,4) but count in the batch statistics)
    ConvParams cp{};
    cp.in = pool1, cp.wfrag = e->w2frag, cp.B = B, cp.nsub = ns;
    cp.in_rows = e->H1, cp.in_cols = L.W1;
    cp.in_cstride = e->H1 * L.W1;
    cp.in_bandoff = 32 * e->H1 * L.W1;
    cp.in_clipstride = (long long)ns * 32 * e->H1 * L.W1;
    cp.out_rows = e->FD, cp.out_cols = L.W2;
    // the last tile row holds at most 2 real output rows (rows 8, 9 of 10 at the default geometry): those go through
    // the 2 x 40-tile strip kernel instead of 8 x 8 tiles that would be 75 % padding
    const int rows_last = e->H1 - 8 * (T.tr2 - 1);
    const bool strip = T.tr2 >= 2 && rows_last <= 2 && !getenv("MST_CONV2_NO_STRIP");
    cp.tiles_r = strip ? T.tr2 - 1 : T.tr2, cp.tiles_c = T.tc2;
    cp.sets_per_band = (B * cp.tiles_r * cp.tiles_c + kConvWaves - 1) / kConvWaves;
    cp.yraw = y2, cp.stats = stats2, cp.bias = e->c2b, cp.raw_rows = e->H1, cp.raw_cols = L.W1;
    cp.acc_tr = T.tr2, cp.acc_tc = T.tc2;
    if (train_fwd16(e)) {   // all tile rows on the f16 kernel (no strip: its padding costs less than a second launch)
      cp.tiles_r = T.tr2;
      cp.sets_per_band = (B * cp.tiles_r * cp.tiles_c + kConvWaves - 1) / kConvWaves;
      const int g = std::min(grid, ns * cp.sets_per_band);
      constexpr size_t lds = (size_t)(kF16Steps * 4 * 2 * 64 + kConvWaves * 2 * 14 * 14) * 16;
      static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
      if (mst::first_use_on_device(attr_set)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_f16x3_kernel<1, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
          err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_f16x3_kernel<3, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return mst::fail(MST_EHIP, "conv2 (f16 train) attribute failed: %s", hipGetErrorString(err));
      }
      cp.f16_scale = reinterpret_cast<const float*>(ws + T.t_f16scale);
      cp.f16_winv = e->f16_winv2;
      if (e->train_f16 == 1) {
        float* ys2 = reinterpret_cast<float*>(ws + T.t_ys2);
        hipLaunchKernelGGL(f16_yscale_kernel, dim3(ns), dim3(64), 0, st, e->w2norm, e->c2b, 64, static_cast<const unsigned*>(nullptr), B,
                           cp.f16_scale, ys2);
        cp.yraw16 = reinterpret_cast<_Float16*>(y2), cp.y_scale = ys2;
      }
      const h16x8* ih = reinterpret_cast<const h16x8*>(ws + T.t_pool1_h16);
      const h16x8* wf2 = reinterpret_cast<const h16x8*>(e->w2frag16);
      if (e->train_f16 == 2)
        hipLaunchKernelGGL((conv2_f16x3_kernel<3, 1>), dim3(g), dim3(kConvThreads), lds, st, cp, ih,
                           reinterpret_cast<const h16x8*>(ws + T.t_pool1_l16), wf2);
      else hipLaunchKernelGGL((conv2_f16x3_kernel<1, 2>), dim3(g), dim3(kConvThreads), lds, st, cp, ih, ih, wf2);
      MST_HIP_CHECK(hipGetLastError());
    } else {
      const int g = std::min(grid, ns * cp.sets_per_band);
      using GEO = ConvGeom<2, 2>;
      const size_t lds = (size_t)(2 * GEO::WBP + kConvWaves * GEO::PATCH) * sizeof(float);
      static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
      if (mst::first_use_on_device(attr_set)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel<2, 2, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return mst::fail(MST_EHIP, "conv2 (train) attribute failed: %s", hipGetErrorString(err));
      }
      hipLaunchKernelGGL((conv_kernel<2, 2, 1>), dim3(g), dim3(kConvThreads), lds, st, cp);
      MST_HIP_CHECK(hipGetLastError());
    }
    if (strip && !train_fwd16(e)) {
      ConvParams sp = cp;
      sp.row_off = 8 * (T.tr2 - 1);
      sp.tiles_r = (rows_last + 1) / 2, sp.tiles_c = (L.W1 + 39) / 40;
      sp.sets_per_band = (B * sp.tiles_r * sp.tiles_c + kConvWaves - 1) / kConvWaves;
      const int g = std::min(grid, ns * sp.sets_per_band);
      using GEO = ConvGeom<4, 2>;
      static_assert(GEO::WBP == ConvGeom<2, 2>::WBP, "the strip kernel streams conv2's forward weight fragments");
      const size_t lds = (size_t)(2 * GEO::WBP + kConvWaves * GEO::PATCH) * sizeof(float);
      static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
      if (mst::first_use_on_device(attr_set)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel<4, 2, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return mst::fail(MST_EHIP, "conv2 strip attribute failed: %s", hipGetErrorString(err));
      }
      hipLaunchKernelGGL((conv_kernel<4, 2, 3>), dim3(g), dim3(kConvThreads), lds, st, sp);
      MST_HIP_CHECK(hipGetLastError());
    }
  }
  }   // run_b
  if (run_c) {
  {
    float2* bnstat = reinterpret_cast<float2*>(ws + T.bn2);
    FoldParams fp{stats2, e->bn2w, e->bn2b, film, aff2, bnstat, cscale * B * e->H1 * L.W1, e->cfg.bn_eps, ns, 64, 64, 128,
                  phase ? reinterpret_cast<const long long*>(stats2 + ns * 64 * 2) : nullptr, (double)e->H1 * L.W1};
    hipLaunchKernelGGL(bn_fold_kernel, dim3(ns, B), dim3(64), 0, st, fp);
    ApplyParams ap{y2, e->H1, aff2, pool_in, nullptr, 1.f, B, ns, T.tr2, T.tc2, e->FD, L.W2, (long long)B * ns * T.tr2 * T.tc2 * 4 * 64,
                   nullptr, nullptr, L.W1};
    if (e->train_f16 == 1) ap.y_scale = reinterpret_cast<const float*>(ws + T.t_ys2);
    hipLaunchKernelGGL((apply_kernel<2, 2>), dim3((unsigned)((ap.units + 255) / 256)), dim3(256), 0, st, ap);
    MST_HIP_CHECK(hipGetLastError());
  }
  if (emb) {   // attention pooling head (emb == NULL: the caller runs its own head on pool_in)
    AttnParams ap{pool_in, e->att0frag, e->att0_b, e->att2_w, e->att2_b, scores, B, e->C, L.W2, e->cfg.attn_hidden};
    const int mtiles = (B * L.W2 + 16 * kAttnMT - 1) / (16 * kAttnMT);
    hipLaunchKernelGGL(attn_scores_kernel, dim3(mtiles), dim3(256), 0, st, ap);
    MST_HIP_CHECK(hipGetLastError());
    float* pooled = reinterpret_cast<float*>(ws + L.pooled);
    PoolParams pp{pool_in, scores, pooled, e->C, L.W2};
    const int slices = (e->C + 127) / 128;
    hipLaunchKernelGGL(attn_pool_kernel, dim3(B, slices), dim3(256), (size_t)((L.W2 + 3) & ~3) * sizeof(float), st, pp);
    MST_HIP_CHECK(hipGetLastError());
    ProjParams pj{pooled, e->projfrag, e->proj_b, emb, B, e->C, e->cfg.embed_dim};
    const int mt_all = (B + 15) / 16;
    if (mt_all <= 1) launch_proj<1>(pj, 1, st);
    else if (mt_all <= 2) launch_proj<2>(pj, 1, st);
    else if (mt_all <= 3) launch_proj<3>(pj, 1, st);
    else if (mt_all <= 5) launch_proj<5>(pj, 1, st);
    else launch_proj<8>(pj, (B + 127) / 128, st);
    MST_HIP_CHECK(hipGetLastError());
  }
  if (taps && taps->film) MST_HIP_CHECK(hipMemcpyAsync(taps->film, film, (size_t)B * ns * 192 * 4, hipMemcpyDeviceToDevice, st));
  if (taps && taps->bn1) MST_HIP_CHECK(hipMemcpyAsync(taps->bn1, ws + T.bn1, (size_t)ns * 32 * 8, hipMemcpyDeviceToDevice, st));
  if (taps && taps->bn2) MST_HIP_CHECK(hipMemcpyAsync(taps->bn2, ws + T.bn2, (size_t)ns * 64 * 8, hipMemcpyDeviceToDevice, st));
  }   // run_c
  return MST_OK;
}

int mst_encoder_train_backward_apply(const mst_encoder* e, int layer, int B, int frames, const float* dpool,
                                     long long dp_clip, long long dp_band, long long dp_ch, float* dy, float* dfilm,
                                     float* dbn, void* workspace, size_t workspace_bytes, void* stream) {
  return mst_encoder_train_backward_apply_phase(e, layer, B, frames, dpool, dp_clip, dp_band, dp_ch, dy, dfilm, dbn, workspace,
                                                workspace_bytes, stream, 0, 1.0);
}

int mst_encoder_train_backward_apply_phase(const mst_encoder* e, int layer, int B, int frames, const float* dpool,
                                           long long dp_clip, long long dp_band, long long dp_ch, float* dy, float* dfilm,
                                           float* dbn, void* workspace, size_t workspace_bytes, void* stream, int phase,
                                           double count_scale) {
  MST_REQUIRE(e && dpool && dfilm && dbn, "mst_encoder_train_backward_apply: NULL argument");
  MST_REQUIRE(phase >= 0 && phase <= 3, "mst_encoder_train_backward_apply: phase %d (0..3)", phase);
  const bool run_1 = phase == 0 || phase == 1, run_2 = phase == 0 || phase == 2, run_3 = phase == 0 || phase == 3;
  const double cscale = count_scale > 0.0 ? count_scale : 1.0;

  MST_REQUIRE((layer == 1 || layer == 2) && B > 0 && frames >= 20, "mst_encoder_train_backward_apply: bad arguments");
  const TrainLayout T = train_layout(e, B, frames);
  const WsLayout& L = T.base;
  if (!workspace || workspace_bytes < T.total)
    return mst::fail(MST_ENOMEM, "mst_encoder_train_backward_apply: workspace %zu B < required %zu B", workspace_bytes, T.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int ns = e->cfg.n_subbands;
  ApplyBwdParams p{};
  p.film = reinterpret_cast<const float*>(ws + L.film);
  p.dpool = dpool, p.dp_clip = dp_clip, p.dp_band = dp_band, p.dp_ch = dp_ch;
  p.dfilm = dfilm, p.dy = dy, p.B = B, p.nsub = ns;
  const int cout = layer == 1 ? 32 : 64;
  mst::DetAcc* sums = reinterpret_cast<mst::DetAcc*>(ws + (layer == 1 ? T.stats1 : T.stats2));   // forward sums are spent: reuse
  p.sums = sums;
  p.dfilm_acc = reinterpret_cast<mst::DetAcc*>(ws + T.dfilm_acc);
  if (run_2) {
    MST_HIP_CHECK(hipMemsetAsync(sums, 0, (size_t)ns * cout * 2 * sizeof(mst::DetAcc), st));
    MST_HIP_CHECK(hipMemsetAsync(p.dfilm_acc, 0, (size_t)B * ns * 192 * sizeof(mst::DetAcc), st));
    if (phase != 0)
      hipLaunchKernelGGL(set_clips_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<long long*>(sums + ns * cout * 2),
                         static_cast<long long*>(nullptr), (long long)B);
  }
  if (phase != 0) p.clips = reinterpret_cast<const long long*>(sums + ns * cout * 2);
  // f16 training: the backward pass runs under an internal power-of-two loss scale s chosen from max |d pool_in| (layer 2 =
  // the start of the trunk's backward); d pool1 carries it to layer 1; every result that leaves the trunk is divided by s.
  // In this mode `dy` of layer 2 is the f16 channel-minor operand of the f16 dgrad kernel, [n_sub][B][H1][W1][64] halves.
  const float* unscale = nullptr;
  if (train_bwd16(e)) {
    float* bs = reinterpret_cast<float*>(ws + T.t_bscale);
    unscale = bs;
    if (layer == 2) {
      MST_REQUIRE(dp_ch == (long long)e->FD * L.W2 && dp_band == 64 * dp_ch && dp_clip == (long long)ns * dp_band,
                  "mst_encoder_train_backward_apply: f16 training expects a contiguous d pool_in");
      unsigned* bits = reinterpret_cast<unsigned*>(bs + 2);
      if (run_1) {
        MST_HIP_CHECK(hipMemsetAsync(bits, 0, sizeof(unsigned), st));
        hipLaunchKernelGGL(absmax_kernel, dim3(256, 1), dim3(256), 0, st, dpool, (long long)B * dp_clip, bits);
      }
      if (run_2) hipLaunchKernelGGL(f16_bscale_kernel, dim3(1), dim3(1), 0, st, bits, bs);
      p.in_scale = bs;
      p.dy_h16 = reinterpret_cast<_Float16*>(dy);
      p.dy = nullptr;
    }
  }
  if (layer == 1) {
    p.yraw = reinterpret_cast<const float*>(ws + T.y1), p.aff = reinterpret_cast<const float2*>(ws + L.aff1);
    p.dy_acc = dy ? nullptr : reinterpret_cast<float*>(ws + T.y1);
    p.bnstat = reinterpret_cast<const float2*>(ws + T.bn1), p.bn_w = e->bn1w, p.bn_b = e->bn1b;
    p.dp_rows = e->H1, p.dp_cols = L.W1, p.tiles_r = T.tr1, p.tiles_c = T.tc1;
    p.pool_h = e->sub;
    if (e->train_f16 == 1) p.y_scale = reinterpret_cast<const float*>(ws + T.t_ys1);
    p.rows = e->cfg.split_size, p.cols = frames, p.goff = 0, p.boff = 32;
    p.count = cscale * B * e->cfg.split_size * frames;
  } else {
    p.yraw = reinterpret_cast<const float*>(ws + T.y2), p.aff = reinterpret_cast<const float2*>(ws + L.aff2);
    p.dy_acc = reinterpret_cast<float*>(ws + T.y2);   // layer 2: always kept in accumulator order as well (conv2 wgrad; fp32 mode only)
    if (e->train_f16 == 1) p.y_scale = reinterpret_cast<const float*>(ws + T.t_ys2);
    p.bnstat = reinterpret_cast<const float2*>(ws + T.bn2), p.bn_w = e->bn2w, p.bn_b = e->bn2b;
    p.dp_rows = e->FD, p.dp_cols = L.W2, p.tiles_r = T.tr2, p.tiles_c = T.tc2;
    p.rows = e->H1, p.cols = L.W1, p.goff = 64, p.boff = 128;
    p.count = cscale * B * e->H1 * L.W1;
  }
  const int nt = layer == 1 ? 2 : 4;
  const int wus = p.tiles_r * p.tiles_c * nt;
  p.chunks = std::max(1, std::min(16, wus / 64));
  const long long units = (long long)B * ns * wus * 64;
  const dim3 gr(p.chunks, B * ns), gd((unsigned)((units + 255) / 256));
  // pass A (run_2): the sums over this rank's clips, and from them the BatchNorm parameter gradients and the FiLM gradients --
  // LOCAL contributions: with cross-rank statistics the caller all-reduces the sums AFTER this point, and its ordinary
  // gradient all-reduce adds up the ranks' dbn.  pass B (run_3): d(conv output) from the (global) sums.
  if (run_2) {
    if (layer == 1 && (e->sub == 2 || train_bwd16(e))) hipLaunchKernelGGL((apply_bwd_reduce_kernel<1, 2>), gr, dim3(256), 0, st, p);
    else if (layer == 1) hipLaunchKernelGGL((apply_bwd_reduce_kernel<1, 1>), gr, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((apply_bwd_reduce_kernel<2, 2>), gr, dim3(256), 0, st, p);
    MST_HIP_CHECK(hipGetLastError());
    // dbn = two planes [band][ch]: dgamma_bn = S2, then dbeta_bn = S1, as fp32
    hipLaunchKernelGGL(sums_to_dbn_kernel, dim3((ns * cout + 255) / 256), dim3(256), 0, st, sums, dbn, ns * cout, unscale);
    hipLaunchKernelGGL(dfilm_finish_kernel, dim3((B * ns * 2 * cout + 255) / 256), dim3(256), 0, st, p.dfilm_acc, dfilm,
                       B * ns * 192, p.goff, p.boff, cout, unscale);
    MST_HIP_CHECK(hipGetLastError());
  }
  if (run_3) {
    if (train_bwd16(e)) {   // f16 d(conv output) in the weight gradient's operand layout
      MST_REQUIRE(layer == 2 || dy == nullptr, "mst_encoder_train_backward_apply: f16 training keeps layer 1's dy in the workspace (pass dy = NULL)");
      const int CG = (B + 7) / 8;
      const long long gunits = (long long)ns * CG * wus * 64;
      const dim3 gg((unsigned)((2 * gunits + 255) / 256));   // two threads (4 clips each) per lane-unit
      const bool x3 = e->train_f16 == 2;
      if (layer == 1) {
        h16x8* dyg = reinterpret_cast<h16x8*>(ws + T.t_dyg1);
        if (x3) hipLaunchKernelGGL((apply_bwd_dx_f16_kernel<1, 2, 3>), gg, dim3(256), 0, st, p, gunits, dyg, CG);
        else hipLaunchKernelGGL((apply_bwd_dx_f16_kernel<1, 2, 1>), gg, dim3(256), 0, st, p, gunits, dyg, CG);
      } else {
        h16x8* dyg = reinterpret_cast<h16x8*>(ws + T.t_dyg2);
        if (x3) hipLaunchKernelGGL((apply_bwd_dx_f16_kernel<2, 2, 3>), gg, dim3(256), 0, st, p, gunits, dyg, CG);
        else hipLaunchKernelGGL((apply_bwd_dx_f16_kernel<2, 2, 1>), gg, dim3(256), 0, st, p, gunits, dyg, CG);
      }
    } else if (layer == 1 && e->sub == 2) {
      hipLaunchKernelGGL((apply_bwd_dx_kernel<1, 2>), gd, dim3(256), 0, st, p, units);
    } else if (layer == 1) {
      hipLaunchKernelGGL((apply_bwd_dx_kernel<1, 1>), gd, dim3(256), 0, st, p, units);
    } else {
      hipLaunchKernelGGL((apply_bwd_dx_kernel<2, 2>), gd, dim3(256), 0, st, p, units);
    }
    MST_HIP_CHECK(hipGetLastError());
  }
  (void)run_1;
  return MST_OK;
}

namespace {
// six small device-to-device copies in one launch (blockIdx.y = which)
struct CopySix {
  float* dst[6];
  const float* src[6];
  int n[6];
};
__global__ void copy_six_kernel(const CopySix c) {
  const int k = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c.n[k]) c.dst[k][i] = c.src[k][i];
}

// running statistics of a BatchNorm layer after a training forward, as nn.BatchNorm2d updates them (src/model.py:107-125 under
// model.train()): running = (1 - m) running + m batch, the batch variance unbiased (count / (count - 1)); stat = (mean, 1/sqrt(biased
// var + eps)) as the training forward left it; clips != NULL: count = clips[0] * per_clip (cross-rank statistics), else `count`
__global__ void bn_running_update_kernel(const float2* __restrict__ stat, float* __restrict__ rmean, float* __restrict__ rvar,
                                         long long* __restrict__ nbatches, int n, int nsub, float momentum, float eps, double count,
                                         const long long* __restrict__ clips, double per_clip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nsub) nbatches[i] += 1;
  if (i >= n) return;
  const double cnt = clips ? (double)clips[0] * per_clip : count;
  const float corr = (float)(cnt / fmax(cnt - 1.0, 1.0));
  const float2 s = stat[i];
  const float var = __fmul_rn(__fsub_rn(__frcp_rn(__fmul_rn(s.y, s.y)), eps), corr);
  rmean[i] = __fadd_rn(__fmul_rn(rmean[i], 1.f - momentum), __fmul_rn(s.x, momentum));
  rvar[i] = __fadd_rn(__fmul_rn(rvar[i], 1.f - momentum), __fmul_rn(var, momentum));
}
}  // namespace

int mst_encoder_train_update_running_stats(const mst_encoder* e, int layer, int B, int frames, float* running_mean, float* running_var,
                                           long long* num_batches_tracked, float momentum, int cross_rank, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  MST_REQUIRE(e && (layer == 1 || layer == 2) && running_mean && running_var && num_batches_tracked && B > 0 && frames >= 20,
              "mst_encoder_train_update_running_stats: bad argument");
  const TrainLayout T = train_layout(e, B, frames);
  if (!workspace || workspace_bytes < T.total)
    return mst::fail(MST_ENOMEM, "mst_encoder_train_update_running_stats: workspace %zu B < required %zu B", workspace_bytes, T.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int ns = e->cfg.n_subbands, cout = layer == 1 ? 32 : 64;
  const double per_clip = layer == 1 ? (double)e->cfg.split_size * frames : (double)e->H1 * T.base.W1;
  const mst::DetAcc* stats = reinterpret_cast<const mst::DetAcc*>(ws + (layer == 1 ? T.stats1 : T.stats2));
  hipLaunchKernelGGL(bn_running_update_kernel, dim3((ns * cout + 255) / 256), dim3(256), 0, st,
                     reinterpret_cast<const float2*>(ws + (layer == 1 ? T.bn1 : T.bn2)), running_mean, running_var, num_batches_tracked,
                     ns * cout, ns, momentum, e->cfg.bn_eps, (double)B * per_clip,
                     cross_rank ? reinterpret_cast<const long long*>(stats + ns * cout * 2) : nullptr, per_clip);
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

int mst_encoder_update_trunk_params(mst_encoder* e, const float* conv1_w, const float* conv1_b, const float* bn1_w,
                                    const float* bn1_b, const float* conv2_w, const float* conv2_b, const float* bn2_w,
                                    const float* bn2_b, void* stream) {
  MST_REQUIRE(e && conv1_w && conv1_b && bn1_w && bn1_b && conv2_w && conv2_b && bn2_w && bn2_b,
              "mst_encoder_update_trunk_params: NULL argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int ns = e->cfg.n_subbands;
  {
    const long long t1 = (long long)ns * 2 * 49 * 2 * 64, t2 = (long long)ns * 8 * 49 * 4 * 64;
    hipLaunchKernelGGL(conv_fragments_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, st, conv1_w, e->w1frag, ns, 32, 8,
                       ConvGeom<1, 2>::WBP);
    hipLaunchKernelGGL(conv_fragments_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, st, conv2_w, e->w2frag, ns, 64, 32,
                       ConvGeom<2, 2>::WBP);
    const long long t3 = (long long)ns * 16 * 49 * 2 * 64;
    hipLaunchKernelGGL(dgrad_fragments_kernel, dim3((unsigned)((t3 + 255) / 256)), dim3(256), 0, st, conv2_w, e->w2dfrag, ns,
                       ConvGeom<3, 2>::WBP);
    MST_HIP_CHECK(hipGetLastError());
  }
  if (train_fwd16(e)) {   // f16 fragments of the new weights: per-channel power-of-two pre-scale, then hi/lo (forward) / hi (dgrad)
    hipLaunchKernelGGL(f16_wstats_kernel, dim3(ns * 32), dim3(64), 0, st, conv1_w, 32, (long long)32 * 392, 392, 1, 0, 392,
                       e->f16_wsc1, e->f16_winv1, e->w1norm);
    hipLaunchKernelGGL(f16_wstats_kernel, dim3(ns * 64), dim3(64), 0, st, conv2_w, 64, (long long)64 * 1568, 1568, 1, 0, 1568,
                       e->f16_wsc2, e->f16_winv2, e->w2norm);
    hipLaunchKernelGGL(f16_wstats_kernel, dim3(ns * 32), dim3(64), 0, st, conv2_w, 32, (long long)64 * 1568, 49, 64, 1568, 49,
                       e->f16_wsc2d, e->f16_winv2d, static_cast<float*>(nullptr));
    const long long n1 = (long long)ns * 1 * kF16Steps * 2 * 512, n2 = (long long)ns * 4 * kF16Steps * 4 * 512,
                    n3 = (long long)ns * 8 * kF16Steps * 2 * 512;
    hipLaunchKernelGGL(f16_fragments_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, conv1_w, e->f16_wsc1,
                       reinterpret_cast<_Float16*>(e->w1frag16), ns, 32, 8, 2, 0);
    hipLaunchKernelGGL(f16_fragments_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, conv2_w, e->f16_wsc2,
                       reinterpret_cast<_Float16*>(e->w2frag16), ns, 64, 32, 2, 0);
    hipLaunchKernelGGL(f16_fragments_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, conv2_w, e->f16_wsc2d,
                       reinterpret_cast<_Float16*>(e->w2dfrag16), ns, 32, 64, e->train_f16 == 2 ? 2 : 1, 1);
    MST_HIP_CHECK(hipGetLastError());
  }
  CopySix c6{{e->c1b, e->bn1w, e->bn1b, e->c2b, e->bn2w, e->bn2b}, {conv1_b, bn1_w, bn1_b, conv2_b, bn2_w, bn2_b},
             {ns * 32, ns * 32, ns * 32, ns * 64, ns * 64, ns * 64}};
  hipLaunchKernelGGL(copy_six_kernel, dim3((ns * 64 + 255) / 256, 6), dim3(256), 0, st, c6);   // one launch instead of six copies
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

// ---- device-side refresh of EVERY table of the encoder (eval and training) from device tensors in state_dict layout
// BatchNorm(eval) folded with the convolution bias: y = s * conv + t  (as mst_encoder_create does on the host, in double)
__global__ void bn_fold_eval_kernel(const float* cb, const float* bw, const float* bb, const float* mean, const float* var, int n,
                                    float eps, float* s, float* t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double sc = (double)bw[i] / sqrt((double)var[i] + (double)eps);
  s[i] = (float)sc;
  t[i] = (float)(sc * ((double)cb[i] - (double)mean[i]) + (double)bb[i]);
}
// [rows][cols] -> [cols][rows]
__global__ void transpose_kernel(const float* __restrict__ w, float* __restrict__ t, int rows, int cols) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * cols) return;
  const int c = (int)(i / rows), r = (int)(i % rows);   // destination-major: coalesced stores
  t[i] = w[(size_t)r * cols + c];
}
// Linear weight [N][K] -> fp32-MFMA B fragments [K/4][N/16][64]: lane (n, kq) of step s, tile nt holds w[16 nt + n][4 s + kq]
__global__ void linear_fragments_kernel(const float* __restrict__ w, float* __restrict__ f, int N, int K) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * K) return;
  const int lane = (int)(i & 63);
  const long long r = i >> 6;
  const int nt = (int)(r % (N / 16)), st = (int)(r / (N / 16));
  f[i] = w[(size_t)(nt * 16 + (lane & 15)) * K + 4 * st + (lane >> 4)];
}
// conv1 weights as float16 hi / lo B fragments in conv1_f16e_kernel's k order: [band][14 steps][2 nt][hi/lo][lane][8 channels],
// step = (tap column st >> 1, upper / lower tap rows st & 1), lane group kq -> tap row 4 (st & 1) + (0, 2, 1, 3)[kq]
__global__ void f16e_fragments_kernel(const float* __restrict__ w, const float* __restrict__ scale, _Float16* f, int nsub) {
  const long long total = (long long)nsub * kF16StepsE * 2 * 64 * 8;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
  long long r = i >> 9;
  const int n = (int)(r % 2);
  r /= 2;
  const int st = (int)(r % kF16StepsE), b = (int)(r / kF16StepsE);
  const int kq = lane >> 4, tr = 4 * (st & 1) + (((kq & 1) << 1) | (kq >> 1)), tc = st >> 1, co = n * 16 + (lane & 15);
  const float v = tr < 7 ? w[(((size_t)b * 32 + co) * 8 + j) * 49 + tr * 7 + tc] * scale[b * 32 + co] : 0.f;
  const _Float16 h = (_Float16)v;
  const size_t base = ((((size_t)b * kF16StepsE + st) * 2 + n) * 2) * 512;
  f[base + (size_t)lane * 8 + j] = h;
  f[base + 512 + (size_t)lane * 8 + j] = (_Float16)(v - (float)h);
}

int mst_encoder_update_params(mst_encoder* e, const mst_encoder_weights* w, void* stream) {
  MST_REQUIRE(e && w, "mst_encoder_update_params: NULL argument");
  for (const float* const* q = &w->conv1_w; q <= &w->proj_b; ++q)
    MST_REQUIRE(*q != nullptr, "mst_encoder_update_params: NULL weight pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int ns = e->cfg.n_subbands, H = e->cfg.film_hidden, Fd = e->cfg.feature_dim, A = e->cfg.attn_hidden, E = e->cfg.embed_dim, C = e->C;
  // trunk: fp32 fragments, un-folded parameters, (training f16 modes) their float16 fragments
  int rc = mst_encoder_update_trunk_params(e, w->conv1_w, w->conv1_b, w->bn1_w, w->bn1_b, w->conv2_w, w->conv2_b, w->bn2_w, w->bn2_b, stream);
  if (rc) return rc;
  auto blocks = [](long long n) { return dim3((unsigned)((n + 255) / 256)); };
  // eval-mode BatchNorm affines
  hipLaunchKernelGGL(bn_fold_eval_kernel, blocks(ns * 32), dim3(256), 0, st, w->conv1_b, w->bn1_w, w->bn1_b, w->bn1_mean, w->bn1_var,
                     ns * 32, e->cfg.bn_eps, e->s1, e->t1);
  hipLaunchKernelGGL(bn_fold_eval_kernel, blocks(ns * 64), dim3(256), 0, st, w->conv2_b, w->bn2_w, w->bn2_b, w->bn2_mean, w->bn2_var,
                     ns * 64, e->cfg.bn_eps, e->s2, e->t2);
  // eval-path float16 fragments (all three orders) with their per-channel power-of-two pre-scales and the filters' L1 norms
  if (!e->f16_wsc1e) {
    bool ok = hipMalloc(&e->f16_wsc1e, (size_t)ns * 32 * 4) == hipSuccess && hipMalloc(&e->f16_wsc2e, (size_t)ns * 64 * 4) == hipSuccess &&
              hipMalloc(&e->w2norm_e, (size_t)ns * 64 * 4) == hipSuccess;
    if (!ok) return mst::fail(MST_ENOMEM, "mst_encoder_update_params: out of device memory");
  }
  hipLaunchKernelGGL(f16_wstats_kernel, dim3(ns * 32), dim3(64), 0, st, w->conv1_w, 32, (long long)32 * 392, 392, 1, 0, 392,
                     e->f16_wsc1e, e->f16_winv1, e->w1norm);
  hipLaunchKernelGGL(f16_wstats_kernel, dim3(ns * 64), dim3(64), 0, st, w->conv2_w, 64, (long long)64 * 1568, 1568, 1, 0, 1568,
                     e->f16_wsc2e, e->f16_winv2, e->w2norm_e);
  const long long n1 = (long long)ns * 1 * kF16Steps * 2 * 512, n2 = (long long)ns * 4 * kF16Steps * 4 * 512,
                  n1e = (long long)ns * kF16StepsE * 2 * 512;
  hipLaunchKernelGGL(f16_fragments_kernel, blocks(n1), dim3(256), 0, st, w->conv1_w, e->f16_wsc1e, reinterpret_cast<_Float16*>(e->w1frag16),
                     ns, 32, 8, 2, 0);
  hipLaunchKernelGGL(f16_fragments_kernel, blocks(n2), dim3(256), 0, st, w->conv2_w, e->f16_wsc2e, reinterpret_cast<_Float16*>(e->w2frag16),
                     ns, 64, 32, 2, 0);
  hipLaunchKernelGGL(f16e_fragments_kernel, blocks(n1e), dim3(256), 0, st, w->conv1_w, e->f16_wsc1e, reinterpret_cast<_Float16*>(e->w1frag16e), ns);
  // FiLM MLP (transposed weights), attention pooling head (fp32-MFMA B fragments), biases
  hipLaunchKernelGGL(transpose_kernel, blocks((long long)H * Fd), dim3(256), 0, st, w->mlp0_w, e->w0t, H, Fd);
  hipLaunchKernelGGL(transpose_kernel, blocks((long long)H * H), dim3(256), 0, st, w->mlp3_w, e->w3t, H, H);
  hipLaunchKernelGGL(transpose_kernel, blocks((long long)ns * 192 * H), dim3(256), 0, st, w->head_w, e->hwt, ns * 192, H);
  hipLaunchKernelGGL(linear_fragments_kernel, blocks((long long)E * C), dim3(256), 0, st, w->proj_w, e->projfrag, E, C);
  hipLaunchKernelGGL(linear_fragments_kernel, blocks((long long)A * C), dim3(256), 0, st, w->att0_w, e->att0frag, A, C);
  MST_HIP_CHECK(hipGetLastError());
  const struct { float* dst; const float* src; int n; } cp[] = {
      {e->b0, w->mlp0_b, H}, {e->b3, w->mlp3_b, H}, {e->hb, w->head_b, ns * 192}, {e->att0_b, w->att0_b, A},
      {e->att2_w, w->att2_w, A}, {e->att2_b, w->att2_b, 1}, {e->proj_b, w->proj_b, E}};
  for (const auto& c : cp) MST_HIP_CHECK(hipMemcpyAsync(c.dst, c.src, (size_t)c.n * 4, hipMemcpyDeviceToDevice, st));
  return MST_OK;
}

int mst_encoder_train_conv1_wgrad(const mst_encoder* e, const float* logmel, int B, int frames, float* dw,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  mst_logmel_in in{};
  in.layout = MST_LOGMEL_REF, in.data = logmel;
  return mst_encoder_train_conv1_wgrad_in(e, &in, B, frames, dw, workspace, workspace_bytes, stream);
}

int mst_encoder_train_conv1_wgrad_in(const mst_encoder* e, const mst_logmel_in* lin, int B, int frames, float* dw,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(e && lin && lin->data && dw, "mst_encoder_train_conv1_wgrad: NULL argument");
  const int lay = lin->layout;
  MST_REQUIRE(mst_encoder_train_layout_supported(e, lay),
              "mst_encoder_train_conv1_wgrad: log-mel layout %d does not fit the training kernels of precision mode %d", lay, e->train_f16);
  MST_REQUIRE(lay != MST_LOGMEL_CM16 || e->train_f16 == 1 || lin->lo, "mst_encoder_train_conv1_wgrad: MST_LOGMEL_CM16 needs the low parts in the split-precision mode");
  MST_REQUIRE(lay == MST_LOGMEL_REF || ((reinterpret_cast<uintptr_t>(lin->data) | reinterpret_cast<uintptr_t>(lin->lo)) & 15) == 0,
              "mst_encoder_train_conv1_wgrad: channel-minor log-mel must be 16-byte aligned");
  const float* logmel = static_cast<const float*>(lin->data);
  MST_REQUIRE(B > 0 && frames >= 20, "mst_encoder_train_conv1_wgrad: bad arguments");
  const TrainLayout T = train_layout(e, B, frames);
  if (!workspace || workspace_bytes < T.total)
    return mst::fail(MST_ENOMEM, "mst_encoder_train_conv1_wgrad: workspace %zu B < required %zu B", workspace_bytes, T.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int ns = e->cfg.n_subbands;
  mst::DetAcc* dwa = reinterpret_cast<mst::DetAcc*>(ws + T.dw_acc);
  const long long ndw = (long long)ns * 32 * 392;
  MST_HIP_CHECK(hipMemsetAsync(dwa, 0, (size_t)ndw * sizeof(mst::DetAcc), st));
  WgradParams wp{logmel, reinterpret_cast<const float*>(ws + T.y1), dwa, B, ns, T.tr1, T.tc1,
                 e->cfg.split_size, frames, e->cfg.n_mels * frames, e->cfg.overlap * frames,
                 (long long)8 * e->cfg.n_mels * frames, 0};
  const long long total = (long long)ns * B * T.tr1 * T.tc1;
  MST_REQUIRE(total < (1LL << 31), "mst_encoder_train_conv1_wgrad: too many tiles");
  // (20-mel sub-bands: the kernel fits two workgroups per CU -- 126 registers, 44 KB of LDS -- and is launched that wide)
  const int g8 = (int)std::min<long long>((long long)e->num_cus * (e->sub == 2 ? 2 : 1), total);
  const float* unscale = nullptr;
  if (train_bwd16(e)) {   // f16 operands, K = positions x 8 clips (encoder_f16train.inc)
    unscale = reinterpret_cast<const float*>(ws + T.t_bscale);
    WgradF16Params fp{logmel, reinterpret_cast<const h16x8*>(ws + T.t_dyg1), dwa, nullptr, B, ns, T.tr1, T.tc1,
                      e->cfg.split_size, frames, e->cfg.n_mels * frames, e->cfg.overlap * frames,
                      (long long)8 * e->cfg.n_mels * frames, lin->lo, e->cfg.n_mels, e->cfg.overlap};
    const long long items = (long long)ns * ((B + 7) / 8) * T.tr1 * T.tc1;
    constexpr size_t lds1 = (size_t)(8 * 8 * 46 + 2 * 20 * 64) * 16, lds3 = (size_t)(2 * 8 * 8 * 46 + 2 * 10 * 2 * 64) * 16;
    static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
    if (mst::first_use_on_device(attr_set)) {
      hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f16_kernel<1, 1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
      if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f16_kernel<1, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f16_kernel<1, 1, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
      if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f16_kernel<1, 3, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      if (err != hipSuccess) return mst::fail(MST_EHIP, "conv1 wgrad (f16) attribute failed: %s", hipGetErrorString(err));
    }
    const dim3 gw((int)std::min<long long>(e->num_cus, items));
    if (lay == MST_LOGMEL_CM16) {
      if (e->train_f16 == 2) hipLaunchKernelGGL((wgrad_f16_kernel<1, 3, 2>), gw, dim3(kConvThreads), lds3, st, fp);
      else hipLaunchKernelGGL((wgrad_f16_kernel<1, 1, 2>), gw, dim3(kConvThreads), lds1, st, fp);
    } else if (e->train_f16 == 2) hipLaunchKernelGGL((wgrad_f16_kernel<1, 3>), gw, dim3(kConvThreads), lds3, st, fp);
    else hipLaunchKernelGGL((wgrad_f16_kernel<1, 1>), gw, dim3(kConvThreads), lds1, st, fp);
  } else if (e->sub == 2) hipLaunchKernelGGL((conv1_wgrad_kernel<2, 8>), dim3(g8), dim3(512), 0, st, wp);
  else hipLaunchKernelGGL((conv1_wgrad_kernel<1, 8>), dim3(g8), dim3(512), 0, st, wp);
  hipLaunchKernelGGL(det_to_float_kernel, dim3((unsigned)((ndw + 255) / 256)), dim3(256), 0, st, dwa, dw, ndw, unscale);
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

int mst_encoder_train_conv2_wgrad(const mst_encoder* e, const float* pool1, int B, int frames, float* dw,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  MST_REQUIRE(e && dw, "mst_encoder_train_conv2_wgrad: NULL argument");
  // pool1 == NULL (float16 training modes only): conv2's operand is taken from the float16 pool1 planes the training forward left
  // in the workspace -- the bits the fp32 tensor would be rounded to -- so the forward need not write the fp32 pool1 at all
  MST_REQUIRE(pool1 || train_bwd16(e), "mst_encoder_train_conv2_wgrad: pool1 == NULL needs a float16 training mode");
  MST_REQUIRE(B > 0 && frames >= 20, "mst_encoder_train_conv2_wgrad: bad arguments");
  const TrainLayout T = train_layout(e, B, frames);
  const WsLayout& L = T.base;
  if (!workspace || workspace_bytes < T.total)
    return mst::fail(MST_ENOMEM, "mst_encoder_train_conv2_wgrad: workspace %zu B < required %zu B", workspace_bytes, T.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int ns = e->cfg.n_subbands;
  mst::DetAcc* dwa = reinterpret_cast<mst::DetAcc*>(ws + T.dw_acc);
  const long long ndw = (long long)ns * 64 * 1568;
  MST_HIP_CHECK(hipMemsetAsync(dwa, 0, (size_t)ndw * sizeof(mst::DetAcc), st));
  WgradParams wp{pool1, reinterpret_cast<const float*>(ws + T.y2), dwa, B, ns, T.tr2, T.tc2,
                 e->H1, L.W1, e->H1 * L.W1, 32 * e->H1 * L.W1, (long long)ns * 32 * e->H1 * L.W1,
                 0};
  const long long total = (long long)ns * B * T.tr2 * T.tc2;
  MST_REQUIRE(total < (1LL << 31), "mst_encoder_train_conv2_wgrad: too many tiles");
  const int g = (int)std::min<long long>(e->num_cus & ~3, 4 * total);   // groups of 4 workgroups (one per input-channel chunk)
  const float* unscale = nullptr;
  if (train_bwd16(e)) {
    unscale = reinterpret_cast<const float*>(ws + T.t_bscale);
    WgradF16Params fp{pool1, reinterpret_cast<const h16x8*>(ws + T.t_dyg2), dwa, reinterpret_cast<const float*>(ws + T.t_f16scale),
                      B, ns, T.tr2, T.tc2, e->H1, L.W1, e->H1 * L.W1, 32 * e->H1 * L.W1, (long long)ns * 32 * e->H1 * L.W1,
                      nullptr, 0, 0};
    if (!pool1) fp.x = ws + T.t_pool1_h16, fp.x_lo = ws + T.t_pool1_l16;   // (in_clipstride = elements per clip in either form)
    const long long items = (long long)ns * ((B + 7) / 8) * T.tr2 * T.tc2;
    constexpr size_t lds1 = (size_t)(8 * 14 * 14 + 4 * 16 * 64) * 16, lds3 = (size_t)(2 * 8 * 14 * 14 + 4 * 8 * 2 * 64) * 16;
    static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
    if (mst::first_use_on_device(attr_set)) {
      hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f16_kernel<2, 1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
      if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f16_kernel<2, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f16_kernel<2, 1, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
      if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f16_kernel<2, 3, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      if (err != hipSuccess) return mst::fail(MST_EHIP, "conv2 wgrad (f16) attribute failed: %s", hipGetErrorString(err));
    }
    const dim3 gw((int)std::min<long long>(e->num_cus & ~3, 4 * items));
    if (!pool1) {
      if (e->train_f16 == 2) hipLaunchKernelGGL((wgrad_f16_kernel<2, 3, 3>), gw, dim3(kConvThreads), lds3, st, fp);
      else hipLaunchKernelGGL((wgrad_f16_kernel<2, 1, 3>), gw, dim3(kConvThreads), lds1, st, fp);
    } else if (e->train_f16 == 2) hipLaunchKernelGGL((wgrad_f16_kernel<2, 3>), gw, dim3(kConvThreads), lds3, st, fp);
    else hipLaunchKernelGGL((wgrad_f16_kernel<2, 1>), gw, dim3(kConvThreads), lds1, st, fp);
  } else {
    hipLaunchKernelGGL(conv2_wgrad_kernel, dim3(g), dim3(kConvThreads), 0, st, wp);
  }
  hipLaunchKernelGGL(det_to_float_kernel, dim3((unsigned)((ndw + 255) / 256)), dim3(256), 0, st, dwa, dw, ndw, unscale);
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

int mst_encoder_train_conv2_dgrad(const mst_encoder* e, const float* dy2, int B, int frames, float* dpool1,
                                  const unsigned char* drop1_mask, float drop1_scale, void* stream) {
  MST_REQUIRE(e && dy2 && dpool1, "mst_encoder_train_conv2_dgrad: NULL argument");
  MST_REQUIRE(B > 0 && frames >= 20, "mst_encoder_train_conv2_dgrad: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int ns = e->cfg.n_subbands, H1 = e->H1, W1 = frames / 5;
  ConvParams cp{};
  cp.in = dy2, cp.wfrag = e->w2dfrag, cp.out = dpool1, cp.B = B, cp.nsub = ns;
  cp.in_rows = H1, cp.in_cols = W1;
  cp.in_cstride = H1 * W1;                       // dy2 is [band][B][64][H1][W1]
  cp.in_bandoff = B * 64 * H1 * W1;
  cp.in_clipstride = (long long)64 * H1 * W1;
  cp.out_rows = H1, cp.out_cols = W1;
  cp.tiles_r = (H1 + 1) / 2, cp.tiles_c = (W1 + 39) / 40;
  cp.sets_per_band = (B * cp.tiles_r * cp.tiles_c + kConvWaves - 1) / kConvWaves;
  cp.raw_rows = H1, cp.raw_cols = W1, cp.mask = drop1_mask, cp.mask_scale = drop1_scale;
  MST_REQUIRE((long long)cp.in_bandoff * ns < (1LL << 31), "mst_encoder_train_conv2_dgrad: batch too large for 32-bit band offsets");
  const int g = std::min(e->num_cus, ns * cp.sets_per_band);
  if (train_bwd16(e)) {   // dy2 holds f16, channel-minor [n_sub][B][H1][W1][64] (mst_encoder_train_backward_apply in this mode)
    constexpr size_t lds1 = (size_t)(kF16Steps * 2 * 64 + kConvWaves * 8 * 46) * 16, lds3 = 2 * lds1;
    static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
    if (mst::first_use_on_device(attr_set)) {
      hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_dgrad_f16_kernel<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
      if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_dgrad_f16_kernel<3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      if (err != hipSuccess) return mst::fail(MST_EHIP, "conv2 dgrad (f16) attribute failed: %s", hipGetErrorString(err));
    }
    cp.f16_winv = e->f16_winv2d;
    const h16x8* dh = reinterpret_cast<const h16x8*>(dy2);
    const h16x8* wf = reinterpret_cast<const h16x8*>(e->w2dfrag16);
    if (e->train_f16 == 2)   // split precision: the low parts follow the high parts, [2][n_sub][B][H1][W1][64]
      hipLaunchKernelGGL(conv2_dgrad_f16_kernel<3>, dim3(g), dim3(kConvThreads), lds3, st, cp, dh, dh + (size_t)ns * B * H1 * W1 * 8, wf);
    else hipLaunchKernelGGL(conv2_dgrad_f16_kernel<1>, dim3(g), dim3(kConvThreads), lds1, st, cp, dh, dh, wf);
    MST_HIP_CHECK(hipGetLastError());
    return MST_OK;
  }
  using GEO = ConvGeom<3, 2>;
  const size_t lds = (size_t)(2 * GEO::WBP + kConvWaves * GEO::PATCH) * sizeof(float);
  static unsigned long long attr_set = 0;   // per-device bit mask: the attribute belongs to the device
  if (mst::first_use_on_device(attr_set)) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel<3, 2, 2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return mst::fail(MST_EHIP, "conv2 dgrad attribute failed: %s", hipGetErrorString(err));
  }
  hipLaunchKernelGGL((conv_kernel<3, 2, 2>), dim3(g), dim3(kConvThreads), lds, st, cp);
  MST_HIP_CHECK(hipGetLastError());
  return MST_OK;
}

}  // extern "C"
