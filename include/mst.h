/*
 * mst.h -- C ABI of libmst.so: MI355X (gfx950) kernels for the contrastive data path of
 * barry-mir/mixing-style-transfer (waveform stems -> [augment] -> STFT -> mel -> 64-d mixing
 * features -> FiLM band-split CNN encoder -> embedding -> InfoNCE).
 *
 * The reference is 100 % Python and has no FFI; the drop-in boundary is its Python call
 * contract.  Each entry point below names the reference code whose
 * arithmetic it replaces (paths relative to the reference repo).  The Python mirror of the
 * reference classes (mixing-style-transfer_amd/{mixing_utils,model,loss,data}.py) binds these
 * through ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; all `dev` pointers are HIP device pointers owned by the caller
 *     (PyTorch), all `host` pointers are ordinary host memory read during the call only;
 *   - every function returns 0 on success, a negative MST_E* code otherwise, and never throws;
 *     mst_last_error() returns a thread-local description of the last failure;
 *   - the library allocates device memory only inside plan/encoder handles (constant tables and
 *     weights); per-call scratch is a caller-provided workspace (query the size first);
 *   - launches are asynchronous on the given `hipStream_t` (passed as void*; NULL = default
 *     stream), never synchronise, and are safe to capture into a hipGraph;
 *   - handles are immutable after creation and may be shared by concurrent streams as long as
 *     each in-flight call has its own workspace.
 */
#ifndef MST_H_
#define MST_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MST_ABI_VERSION 1

#define MST_OK 0
#define MST_EINVAL (-1)   /* bad argument (shape, NULL pointer, unsupported n_fft ...) */
#define MST_ENOMEM (-2)   /* workspace too small / device allocation failed          */
#define MST_EHIP (-3)     /* a HIP runtime call or kernel launch failed               */

int mst_version(void);
const char* mst_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Stage A: STFT -> mel -> log-mel + mixing features.
 * Replaces: torchaudio MelSpectrogram as called at src/mixing_utils.py:159,280 and
 * src/model.py:58-65; MixingFeatureExtractor.extract_all_features src/mixing_utils.py:71-105
 * (extract_dynamics :107-139, extract_spectral :141-236, extract_stereo :238-268,
 * extract_masking :270-309, compute_loudness :311-318, _flatten_features :320-357).
 * ------------------------------------------------------------------------------------------ */
typedef struct mst_plan mst_plan;

/* window: host [n_fft] analysis window (periodic Hann in the reference);
 * fb:     host [n_fft/2+1][n_mels] row-major mel filterbank exactly as the host framework built
 *         it (torchaudio melscale_fbanks, fp32); each mel band must have contiguous support.
 * detailed_bins: 0 = reference default (5 spectral features/stem, feature_dim 64);
 *         n>0 = use_detailed_spectral mode with n_spectral_bins=n (feature_dim 4*(9+n+2)+8).
 * n_fft in {512,1024,2048}; n_mels <= 256; hop >= 1.                                    */
int mst_plan_create(mst_plan** out, int sample_rate, int n_fft, int hop, int n_mels,
                    const float* window, const float* fb, int detailed_bins);
void mst_plan_destroy(mst_plan* plan);
int mst_plan_frames(const mst_plan* plan, int T);      /* 1 + T / hop (center=True)            */
int mst_plan_feature_dim(const mst_plan* plan);
size_t mst_melfeat_workspace_bytes(const mst_plan* plan, int B, int T);

/* stems:  dev [B][8][T] fp32, channel order vocals L,R, bass L,R, drums L,R, other L,R.
 * logmel: dev [B][8][n_mels][frames] fp32 = log(mel + 1e-10)  (may be NULL: features only)
 * feats:  dev [B][feature_dim] fp32 in the reference's sorted-key layout (may be NULL)
 * Requires T > n_fft/2 (reflect padding), same as torch.stft.                               */
int mst_melfeat_forward(const mst_plan* plan, const float* stems, int B, int T, float* logmel,
                        float* feats, void* workspace, size_t workspace_bytes, void* stream);
/* Same, for the four separate tensors that baseline_collate_fn (src/data.py:291-328) returns:
 * stems4[s] = dev [B][2][T] for s = vocals, bass, drums, other; clip_stride = floats between consecutive
 * clips of one stem (2*T for contiguous (B,2,T) tensors, 8*T for views of a packed [B][8][T] tensor). */
int mst_melfeat_forward_stems(const mst_plan* plan, const float* const stems4[4], long long clip_stride,
                              int B, int T, float* logmel, float* feats, void* workspace,
                              size_t workspace_bytes, void* stream);
/* Same two entry points for int16 PCM stems (the ingest format of SURVEY.md section 8 f2: pre-decoded PCM
 * shards, half the PCIe and HBM bytes of fp32; the reference decodes `{stem}.mp3` to float at
 * src/data.py:169-199).  Samples are converted in-kernel as float(s) * 2^-15 (exact), so the outputs are
 * bit-identical to the fp32 entry points run on that conversion.  clip_stride is in samples.            */
int mst_melfeat_forward_pcm16(const mst_plan* plan, const int16_t* stems, int B, int T, float* logmel,
                              float* feats, void* workspace, size_t workspace_bytes, void* stream);
int mst_melfeat_forward_stems_pcm16(const mst_plan* plan, const int16_t* const stems4[4], long long clip_stride,
                                    int B, int T, float* logmel, float* feats, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* Log-mel layouts.  MST_LOGMEL_REF is the reference's tensor (what MelSpectrogramPreprocessor.forward returns,
 * src/model.py:41-67).  The two channel-minor layouts are the ENCODER-INTERNAL forms: conv1 reads 8-channel patches, and a
 * (frame, band) holding its 8 channels contiguously (32 bytes fp32; 16 + 16 bytes as float16 high / low parts, x = hi + lo to
 * 22 significant bits) lets stage A write whole 128-byte lines once and conv1 load 256-byte runs per frame; the float16
 * form is what the f16 / split-precision convolutions (mst_encoder_set_precision) consume without any conversion pass.  */
#define MST_LOGMEL_REF 0   /* dev [B][8][n_mels][frames] fp32                                            */
#define MST_LOGMEL_CM32 1  /* dev [B][frames][n_mels][8] fp32, 16-byte aligned                           */
#define MST_LOGMEL_CM16 2  /* dev [B][frames][n_mels][8] float16, twice: high parts and low parts        */
/* 1 if stage A writes `layout` directly for this plan (the sliding-window kernels: n_fft 1024 / hop 256 and n_fft 2048 /
 * hop 512), else 0: ask for MST_LOGMEL_REF then.                                                                      */
int mst_plan_layout_supported(const mst_plan* plan, int layout);
typedef struct mst_melfeat_io {
  const void* stems4[4];  /* dev, per stem [B][2][T]: fp32, or int16 PCM when pcm16 != 0 (as the entry points above) */
  long long clip_stride;  /* samples between consecutive clips of one stem                                          */
  int32_t pcm16;
  int32_t layout;         /* MST_LOGMEL_*: layout of `logmel`                                                        */
  void* logmel;           /* dev, may be NULL (features only)                                                        */
  void* logmel_lo;        /* dev, MST_LOGMEL_CM16 only: the low parts; NULL = write the high parts alone (plain f16) */
  uint32_t* absmax;       /* optional dev [B]: max |log-mel| of every clip as float bits (zeroed by the call) -- the    */
                          /* range bound of the float16 convolutions, so that no separate pass reads the log-mel      */
  float* feats;           /* dev [B][feature_dim], may be NULL                                                       */
} mst_melfeat_io;
/* The same launch as mst_melfeat_forward_stems[_pcm16] with the log-mel in any of the layouts above.                  */
int mst_melfeat_forward_io(const mst_plan* plan, const mst_melfeat_io* io, int B, int T, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stage B: FiLM MLP + band-split Conv2D/BN/FiLM/ReLU/MaxPool x2 + attention pooling (eval).
 * Replaces: MixingFeatureEncoder.forward src/model.py:410-464, SubSpectrogramCNN.forward
 * :127-157 (x n_subbands, loop :345-362), concat/view :332-367, AttentionPooling.forward
 * :187-211.  BatchNorm uses running statistics, Dropout is identity (model.eval()).
 * ------------------------------------------------------------------------------------------ */
typedef struct mst_encoder mst_encoder;

typedef struct mst_encoder_config {
  int32_t n_mels, split_size, overlap, n_subbands;
  int32_t feature_dim, embed_dim;
  int32_t film_hidden;  /* 256: feature_mlp width        (src/model.py:385-408) */
  int32_t attn_hidden;  /* 256: attention hidden width   (src/model.py:284-288) */
  float bn_eps;         /* 1e-5                                                   */
} mst_encoder_config;

/* All pointers: host fp32, tensors exactly as in the reference state_dict.                   */
typedef struct mst_encoder_weights {
  /* per sub-band i: audio_encoder.subnet_cnns.{i}.* , concatenated over i                    */
  const float* conv1_w;  /* [n_sub][32][8][7][7]  */
  const float* conv1_b;  /* [n_sub][32]           */
  const float* bn1_w, *bn1_b, *bn1_mean, *bn1_var; /* [n_sub][32] */
  const float* conv2_w;  /* [n_sub][64][32][7][7] */
  const float* conv2_b;  /* [n_sub][64]           */
  const float* bn2_w, *bn2_b, *bn2_mean, *bn2_var; /* [n_sub][64] */
  /* film_encoder.* */
  const float* mlp0_w, *mlp0_b;   /* feature_mlp.0  [H][feature_dim], [H] */
  const float* mlp3_w, *mlp3_b;   /* feature_mlp.3  [H][H], [H]           */
  const float* head_w, *head_b;   /* film_head      [n_sub*192][H], [n_sub*192] */
  /* audio_encoder.attention_pooling.* ; C = 64*n_sub*freq_dim */
  const float* att0_w, *att0_b;   /* attention.0   [A][C], [A] */
  const float* att2_w, *att2_b;   /* attention.2   [1][A], [1] */
  const float* proj_w, *proj_b;   /* projection.0  [E][C], [E] */
} mst_encoder_weights;

/* Optional device outputs for intermediate activations (parity tests); any may be NULL.      */
typedef struct mst_encoder_taps {
  float* film;     /* dev [B][n_sub*192]                      */
  float* pool1;    /* dev [B][n_sub][32][H1][W1]              */
  float* pool_in;  /* dev [B][64*n_sub*freq_dim][W2]  (input of attention pooling) */
  void* events[6]; /* optional hipEvent_t recorded on the stream: [0] start, [1] after FiLM MLP, [2] after conv1,
                      [3] after conv2, [4] after attention scores, [5] end (per-kernel timing for bench.py) */
} mst_encoder_taps;

int mst_encoder_create(mst_encoder** out, const mst_encoder_config* cfg,
                       const mst_encoder_weights* w);
void mst_encoder_destroy(mst_encoder* enc);
/* Optional: mode 1 runs conv1, mode 2 conv1 and conv2, on the f16 matrix cores with 3-term split precision
 * (x = xh + xl, w = wh + wl; xh*wh + xh*wl + xl*wh, fp32 accumulate; per-product error ~2^-22).  conv2's f16 input is
 * range-scaled per (clip, sub-band) by an exact power of two derived on the device from a bound on conv1's output
 * (||w||_1 * max|log-mel| through the folded affine), so no activation magnitude overflows f16 and nothing is refused.  Mode 3 runs both convolutions with plain f16 operands (xh * wh only, fp32 accumulate):
 * the arithmetic of the reference's `--use_amp` autocast convolutions (src/train.py:251-253), ~1e-3 of fp32.
 * 0 (default) = exact fp32 MFMA.  Call before mst_encoder_workspace_bytes. */
int mst_encoder_set_precision(mst_encoder* enc, int conv1_f16x3);
size_t mst_encoder_workspace_bytes(const mst_encoder* enc, int B, int frames);
/* logmel: dev [B][8][n_mels][frames]; feats: dev [B][feature_dim]; emb: dev [B][embed_dim]  */
int mst_encoder_forward(const mst_encoder* enc, const float* logmel, int frames, const float* feats,
                        int B, float* emb, const mst_encoder_taps* taps, void* workspace,
                        size_t workspace_bytes, void* stream);

/* The same forward with the log-mel in one of the encoder-internal channel-minor layouts that stage A writes directly
 * (MST_LOGMEL_*, mst_melfeat_forward_io): MST_LOGMEL_CM32 for the exact-fp32 conv1 (20-mel sub-bands, or an even
 * split_size below 20), MST_LOGMEL_CM16 (high + low float16 planes; the plain-f16 mode 3 reads the high parts only) for
 * the f16 / split-precision modes of mst_encoder_set_precision.  mst_encoder_layout_supported answers for the encoder's
 * CURRENT precision mode.  absmax: stage A's per-clip max |log-mel| (float bits) -- required with MST_LOGMEL_CM16 in the
 * modes that run conv2 on float16 as well (2, 3), optional otherwise (it saves the pass that derives it).          */
typedef struct mst_logmel_in {
  int32_t layout;          /* MST_LOGMEL_*                                     */
  int32_t pad_;
  const void* data;        /* dev: the log-mel (MST_LOGMEL_CM16: high parts)   */
  const void* lo;          /* dev: MST_LOGMEL_CM16 low parts, else NULL        */
  const uint32_t* absmax;  /* dev [B] or NULL                                  */
} mst_logmel_in;
int mst_encoder_layout_supported(const mst_encoder* enc, int layout);
int mst_encoder_forward_in(const mst_encoder* enc, const mst_logmel_in* logmel, int frames, const float* feats,
                           int B, float* emb, const mst_encoder_taps* taps, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Training forward (first half): the same network with train-mode BatchNorm -- batch statistics over
 * (B, H, W) per (sub-band, channel), biased variance, as nn.BatchNorm2d in training mode (src/model.py:107-125 under
 * model.train()) -- and Dropout as identity (pass p = 0 modules; dropout masks are the caller's business until the
 * backward kernels land).  The raw convolution outputs are kept in the workspace in accumulator order for the
 * backward pass.  taps (all optional, device): film / pool1 / pool_in as above; bn1 [n_sub][32][2], bn2 [n_sub][64][2]
 * = (batch mean, 1/sqrt(biased var + eps)) -- the caller updates its running statistics from them.
 * Needs the default 20-mel sub-bands.  Workspace: mst_encoder_train_workspace_bytes (3.6 GB + 1.1 GB of saved
 * activations at 72 clips of 10 s).                                                                           */
typedef struct mst_encoder_train_taps {
  float* film;
  float* pool1;
  float* pool_in;
  float* bn1;
  float* bn2;
  const float* film_in; /* optional INPUT dev [B][n_sub*192]: FiLM parameters from the caller's own MLP (then feats may
                           be NULL); emb may be NULL as well: stop at pool_in, the caller runs its own pooling head */
  const unsigned char* drop1_mask; /* optional INPUT dev, layout of pool1: Dropout keep-mask after the first pooling
                                      (src/model.py:118: Dropout(0.3)); pool1 = mask ? pooled * drop1_scale : 0 */
  float drop1_scale;               /* 1 / (1 - p) */
  /* Data-parallel training with CROSS-RANK BatchNorm statistics (what the single-process reference computes on
   * the whole batch, src/model.py:107-125 under src/train.py:211).  phase 0 (default): the whole forward.  Otherwise the
   * forward runs in three calls with the same arguments -- 1: FiLM + conv1 raw output and its statistics; 2: BatchNorm 1 +
   * FiLM + pooling + conv2 raw output and its statistics; 3: BatchNorm 2 + FiLM + pooling + head -- and between the calls
   * the caller SUMS the statistics accumulators over its ranks (mst_encoder_train_stats_buffer: 64-bit integers, so the sum
   * is exact and order-independent).  The buffer's last word pair carries the CLIP COUNT: every phased call writes this
   * rank's B there, the all-reduce turns it into the global count, and the BatchNorm kernels divide by it -- ranks may hold
   * different numbers of clips (a ragged last batch).  count_scale: kept for ABI compatibility, ignored by the phased calls
   * (phase 0 normalises by B).                                                                                            */
  int phase;
  double count_scale;
  /* Dropout drawn BY the kernel (instead of drop1_mask): with drop1_mask_out != NULL and 0 < drop1_p < 1 the first-pooling
   * epilogue keeps element o iff philox2x32-10(drop1_seed, o) >= drop1_p * 2^32, scales the kept ones by 1 / (1 - p) and
   * writes the keep-mask (uint8, layout of pool1) to drop1_mask_out -- pass that buffer to mst_encoder_train_conv2_dgrad.  */
  unsigned char* drop1_mask_out;
  uint64_t drop1_seed;
  float drop1_p;
} mst_encoder_train_taps;
size_t mst_encoder_train_workspace_bytes(const mst_encoder* enc, int B, int frames);
/* Where the statistics accumulators of conv layer 1 or 2 sit inside the training workspace: byte offset and number of
 * int64 words ([n_sub][C][2 sums][2 words], then 2 words whose first is the rank's clip count -- sum them all).  The forward (sum y, sum y^2) and the backward pass (sum dz, sum dz * zhat)
 * use the same words; all-reduce them with SUM between the phases.  Returns 0 on success.                          */
int mst_encoder_train_stats_buffer(const mst_encoder* enc, int layer, int B, int frames, size_t* offset_bytes,
                                   size_t* n_int64);
int mst_encoder_forward_train(const mst_encoder* enc, const float* logmel, int frames, const float* feats, int B,
                              float* emb, const mst_encoder_train_taps* taps, void* workspace,
                              size_t workspace_bytes, void* stream);
/* The same with a described log-mel (mst_logmel_in, above).  The float16 training modes (mst_encoder_set_train_precision 1, 2)
 * read stage A's MST_LOGMEL_CM16 planes directly -- no fp32 -> float16 conversion while staging, no pass for the per-clip
 * maximum (absmax is required with that layout); the fp32 mode takes MST_LOGMEL_REF only.
 * mst_encoder_train_layout_supported answers for the CURRENT training precision.  The weight gradient of conv1
 * (mst_encoder_train_conv1_wgrad_in) must be given the same log-mel.                                                  */
int mst_encoder_train_layout_supported(const mst_encoder* enc, int layout);
int mst_encoder_forward_train_in(const mst_encoder* enc, const mst_logmel_in* logmel, int frames, const float* feats, int B,
                                 float* emb, const mst_encoder_train_taps* taps, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* Operand precision of the TRAINING kernels (all six entry points below and mst_encoder_forward_train).
 * 0 (default): exact fp32 MFMA.  1: float16 operands, fp32 accumulation -- the arithmetic of the reference's `--use_amp`
 * step (autocast + GradScaler, src/train.py:251-262, src/params.py:71; BASELINE configs[4]): both operands of every
 * convolution-shaped product are rounded to f16 (forward y = conv(f16 x, f16 w); input gradient conv^T(f16 dy, f16 w);
 * weight gradient corr(f16 x, f16 dy)) and the convolution outputs are STORED as f16 (as autocast's are; BatchNorm's batch
 * statistics are those of the stored values); the BatchNorm / FiLM / pooling arithmetic, their backward, all reductions and
 * the master weights stay fp32.  Every rounded tensor is first multiplied by an exact power of two chosen on the device (weights per
 * output channel; pool1 and the stored conv outputs per band from bounds on their values; the gradients by one factor per backward pass from
 * max |d pool_in| -- the internal equivalent of the reference's loss scale) and the factor is divided out in fp32, so neither
 * f16's ceiling nor its subnormals are reached; no host synchronisation.  Needs 20-mel sub-bands.
 * 2: the same kernels with THREE-TERM SPLIT PRECISION -- every operand is a pair hi + lo of float16 (22 significant bits),
 * a product is lo*hi + hi*lo + hi*hi in three MFMAs, fp32 accumulation: results equivalent to the fp32 kernels (tested at
 * their bar) at a third of the f16 matrix rate, i.e. ~5x the fp32 rate.
 * In modes 1 and 2 the `dy` of mst_encoder_train_backward_apply(layer 2) and the `dy2` of mst_encoder_train_conv2_dgrad are
 * a float16 buffer [n_sub][B][H1][W1][64] (channel-minor; mode 2: [2][n_sub][B][H1][W1][64], low parts second; pass it
 * through), layer 1's dy must be NULL, and d pool_in must be contiguous.
 * Call before mst_encoder_train_workspace_bytes and mst_encoder_update_trunk_params.                              */
int mst_encoder_set_train_precision(mst_encoder* enc, int mode);

/* Refresh the convolution / BatchNorm parameters of the TRAINING kernels from device tensors (concatenated over the
 * sub-bands, reference state_dict shapes) -- once per optimizer step; re-swizzles the MFMA weight fragments on the
 * device.  The eval-mode constants (folded running statistics, f16 fragments) are NOT refreshed: create a new
 * encoder for evaluation after training.                                                                         */
int mst_encoder_update_trunk_params(mst_encoder* enc, const float* conv1_w, const float* conv1_b, const float* bn1_w,
                                    const float* bn1_b, const float* conv2_w, const float* conv2_b,
                                    const float* bn2_w, const float* bn2_b, void* stream);

/* Running statistics of BatchNorm layer 1 / 2 after mst_encoder_forward_train, as nn.BatchNorm2d updates them in training mode
 * (src/model.py:107-125): running = (1 - momentum) running + momentum batch (unbiased batch variance), num_batches_tracked += 1.
 * running_mean / running_var: dev [n_sub][C] (the sub-bands' buffers stacked), num_batches_tracked: dev int64 [n_sub];
 * reads the batch statistics the forward left in `workspace` (same buffer, B, frames); cross_rank != 0: the element count is taken
 * from the summed clip-count word of the statistics buffer (mst_encoder_train_stats_buffer), i.e. the GLOBAL batch.  One launch. */
int mst_encoder_train_update_running_stats(const mst_encoder* enc, int layer, int B, int frames, float* running_mean,
                                           float* running_var, long long* num_batches_tracked, float momentum, int cross_rank,
                                           void* workspace, size_t workspace_bytes, void* stream);

/* Device-side refresh of EVERY parameter table of an existing encoder -- what a trainer calls before a validation pass after
 * optimizer steps (src/train.py:388-427 follows :292-296): `w` has mst_encoder_create's fields, but every pointer is a DEVICE
 * tensor in state_dict layout (the sub-band tensors stacked over the bands).  Fragment swizzles, the eval BatchNorm fold, the
 * float16 fragments with their power-of-two pre-scales, transposes and copies all run as kernels on `stream`: no host
 * round trip, no synchronisation, no allocation after the first call.  The configuration (shapes) must be the one the
 * encoder was created with.                                                                                           */
int mst_encoder_update_params(mst_encoder* enc, const mst_encoder_weights* device_weights, void* stream);

/* Backward of max-pool / ReLU / FiLM / BatchNorm(batch statistics) of conv layer 1 or 2, from the activations that
 * mst_encoder_forward_train left in `workspace` (same buffer, same B and frames).
 * dpool: gradient of the pooled activation; element (clip, band, ch, r, c) at
 *        dpool[clip*dp_clip + band*dp_band + ch*dp_ch + r*cols + c]   (layer 1: pool1 10 x W1; layer 2: pool_in 2 x W2).
 * dy:    out, gradient of the convolution output, [n_sub][B][C][rows][cols] (per band a contiguous NCHW tensor: the
 *        operand of a library convolution weight / input gradient); layer 1 only: NULL = keep it in the workspace in
 *        accumulator order for mst_encoder_train_conv1_wgrad.
 * dfilm: [B][n_sub*192], the layer's gamma / beta slots are ACCUMULATED (+=): zero it once per step.
 * dbn:   out [2][n_sub][C]: the plane of d BatchNorm weight, then the plane of d BatchNorm bias (each is the stacked
 *        gradient tensor of its parameter family).                                                               */
int mst_encoder_train_backward_apply(const mst_encoder* enc, int layer, int B, int frames, const float* dpool,
                                     long long dp_clip, long long dp_band, long long dp_ch, float* dy, float* dfilm,
                                     float* dbn, void* workspace, size_t workspace_bytes, void* stream);
/* The same in phases, for cross-rank BatchNorm statistics (see mst_encoder_train_taps::phase): 0 = everything (the call
 * above); 1 = (f16 training modes, layer 2) max |d pool_in| of this rank into the scale word -- all-reduce it with MAX
 * (mst_encoder_train_scale_buffer), so that every rank derives the same internal loss scale; 2 = pass A, the sums over this
 * rank's clips -- all-reduce the layer's statistics buffer with SUM; 3 = pass B and the outputs.  count_scale as above.  */
int mst_encoder_train_backward_apply_phase(const mst_encoder* enc, int layer, int B, int frames, const float* dpool,
                                           long long dp_clip, long long dp_band, long long dp_ch, float* dy, float* dfilm,
                                           float* dbn, void* workspace, size_t workspace_bytes, void* stream, int phase,
                                           double count_scale);
/* f16 training modes: byte offset inside the workspace of the 32-bit word that holds max |d pool_in| as a float bit pattern
 * (non-negative floats order like unsigned integers: all-reduce with MAX on int32).  Returns non-zero in fp32 mode.     */
int mst_encoder_train_scale_buffer(const mst_encoder* enc, int B, int frames, size_t* offset_bytes);

/* conv1 weight gradient, hand-written fp32-MFMA GEMM over positions.  Call after
 * mst_encoder_train_backward_apply(layer 1) with dy == NULL: that variant leaves d(conv1 output) in the workspace, in
 * place of the saved activation, in accumulator order (the operand layout of this kernel).
 * dw: out dev [n_sub][32][8][7][7] (zeroed here).  The bias gradient of a convolution that feeds a batch-statistics
 * BatchNorm is identically zero and is not computed.                                                            */
int mst_encoder_train_conv1_wgrad(const mst_encoder* enc, const float* logmel, int B, int frames, float* dw,
                                  void* workspace, size_t workspace_bytes, void* stream);
int mst_encoder_train_conv1_wgrad_in(const mst_encoder* enc, const mst_logmel_in* logmel, int B, int frames, float* dw,
                                     void* workspace, size_t workspace_bytes, void* stream);

/* conv2 weight gradient, same scheme (dy of layer 2 is always left in the workspace in accumulator order by
 * mst_encoder_train_backward_apply(layer 2), next to the NCHW copy it returns).
 * pool1: dev [B][n_sub][32][H1][W1], the (dropped-out) input of conv2 as the training forward produced it.
 *        NULL in the float16 training modes: the operand is taken from the float16 pool1 planes the training forward left in
 *        the workspace (the bits the fp32 tensor would be rounded to); mst_encoder_forward_train then writes the fp32 pool1
 *        only when taps->pool1 asks for it.
 * dw:    out dev [n_sub][64][32][7][7] (zeroed here).                                                           */
int mst_encoder_train_conv2_wgrad(const mst_encoder* enc, const float* pool1, int B, int frames, float* dw,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* conv2 input gradient: the chunked fp32-MFMA convolution kernel run on d(conv2 output) with the transposed, flipped
 * weights (64 -> 32 channels), Dropout keep-mask applied on the way out.
 * dy2:    dev [n_sub][B][64][H1][W1]  (the NCHW-per-band tensor mst_encoder_train_backward_apply(layer 2) returns).
 * dpool1: out dev [B][n_sub][32][H1][W1] = gradient of pool1 (feed it to backward_apply(layer 1)).
 * drop1_mask / drop1_scale: as in mst_encoder_train_taps (NULL: no dropout).                                      */
int mst_encoder_train_conv2_dgrad(const mst_encoder* enc, const float* dy2, int B, int frames, float* dpool1,
                                  const unsigned char* drop1_mask, float drop1_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Augmentation chain.  Replaces AudioAugmenter.augment_stems src/mixing_utils.py:376-419 and
 * apply_spectral_tilt :421-433, apply_compression :435-447, apply_bandwidth_limit :449-456,
 * apply_reverb :458-479.  Every random decision is drawn by the HOST in the reference's order
 * with the reference's RNG (torch CPU generator) and passed in; the kernels are deterministic.
 * ------------------------------------------------------------------------------------------ */
typedef struct mst_aug_stem {
  float gain;          /* linear gain, 1.0f = none             (:389-392)                   */
  int32_t tilt;        /* 0 none, 1 section in tilt_sos        (:421-433)                   */
  int32_t compress;    /* 0 none; 1: the reference's default 4:1 above -20 dB; 2: comp_threshold_db / comp_ratio (:435-447) */
  int32_t bw_sections; /* 0 none, else #biquads in bw_sos (2)  (:449-456)                   */
  double tilt_sos[6];  /* scipy.signal.butter(..., output='sos') rows b0 b1 b2 a0 a1 a2     */
  double bw_sos[12];
  float comp_threshold_db; /* compress == 2: apply_compression(audio, threshold, ratio), threshold in dB ... */
  float comp_ratio;        /* ... and ratio > 0                                               */
} mst_aug_stem;

typedef struct mst_aug_clip {
  mst_aug_stem stem[4];
  int32_t reverb;      /* 0/1: reverb of the summed mix, redistributed (:404-417)           */
  int32_t pad_;
} mst_aug_clip;

size_t mst_aug_workspace_bytes(int B, int T, int ir_len);
/* stems_inout: dev [B][8][T] modified in place; decisions: host [B];
 * reverb_ir:   dev [B][ir_len] (rows of clips with reverb==0 are ignored; may be NULL if none) */
int mst_aug_apply(const mst_aug_clip* decisions, int B, int T, float* stems_inout,
                  const float* reverb_ir, int ir_len, void* workspace, size_t workspace_bytes,
                  void* stream);

/* The same on clips that sit `clip_stride` floats apart (>= 8 * T; each clip's 8 channels contiguous): e.g. the
 * negatives of a triplet batch, every third clip of the batch tensor, augmented where they stand.               */
int mst_aug_apply_strided(const mst_aug_clip* decisions, int B, int T, float* stems_inout, long long clip_stride,
                          const float* reverb_ir, int ir_len, void* workspace, size_t workspace_bytes,
                          void* stream);

/* Out of place: stems_out[b] = augment(src[b]) -- the reference's `stems.clone()` (src/mixing_utils.py:386) folded into the
 * first pass over the audio instead of a copy before it.  src: dev, B clips `src_stride` floats apart, never written; stems_out:
 * dev, B clips `clip_stride` floats apart, every sample written (a stem without a decision is copied).  The address ranges of
 * src and stems_out must be disjoint (MST_EINVAL otherwise); src == stems_out with equal strides is mst_aug_apply_strided.  */
int mst_aug_apply_from(const mst_aug_clip* decisions, int B, int T, const float* src, long long src_stride,
                       float* stems_out, long long clip_stride, const float* reverb_ir, int ir_len,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training forward / backward of the two small networks around the convolution trunk, fp32, deterministic
 * (fixed-order reductions, no atomics), Dropout masks never stored -- a keep decision is a pure function of
 * (seed, element index), Philox-2x32-10, re-derived by every kernel that needs it; p = 0 switches a Dropout off.
 * All weight / gradient pointers are DEVICE tensors in state_dict layout (row-major nn.Linear weights [out][in]).
 *
 * Attention-pooling head = the reference's  F.dropout -> AttentionPooling  (src/model.py:118 for the Dropout in front,
 * :187-211):  xd = Dropout_in(pool_in);  a = softmax_t(att2 . tanh(att0 xd_t + b0) + b2);  pooled = sum_t a_t xd_t;
 * emb = Dropout_out(ReLU(proj pooled + bp)).     pool_in: dev [B][channels][frames];  emb: dev [B][embed_dim].
 * `save` (mst_head_save_bytes) carries the activations the backward needs and must stay untouched until then.
 * mst_head_backward: demb [B][embed_dim] -> the six parameter gradients (overwritten) and dpool_in [B][channels][frames].
 * ------------------------------------------------------------------------------------------ */
typedef struct mst_head_dims { int32_t channels, frames, attn_hidden, embed_dim; } mst_head_dims;   /* attn_hidden <= 256, channels <= 3072 */
typedef struct mst_head_weights {
  const float *att0_w, *att0_b;   /* attention.0   [A][C], [A] */
  const float *att2_w, *att2_b;   /* attention.2   [1][A], [1] */
  const float *proj_w, *proj_b;   /* projection.0  [E][C], [E] */
} mst_head_weights;
typedef struct mst_head_grads { float *att0_w, *att0_b, *att2_w, *att2_b, *proj_w, *proj_b; } mst_head_grads;
size_t mst_head_save_bytes(const mst_head_dims* dims, int B, float drop_in_p);   /* (holds Dropout_in(pool_in) when drop_in_p > 0) */
size_t mst_head_backward_workspace_bytes(const mst_head_dims* dims, int B);
int mst_head_forward_train(const mst_head_dims* dims, const mst_head_weights* w, const float* pool_in, int B, float drop_in_p,
                           uint64_t drop_in_seed, float drop_out_p, uint64_t drop_out_seed, float* emb, void* save,
                           size_t save_bytes, void* stream);
int mst_head_backward(const mst_head_dims* dims, const mst_head_weights* w, const float* pool_in, int B, float drop_in_p,
                      uint64_t drop_in_seed, float drop_out_p, uint64_t drop_out_seed, const float* demb, const void* save,
                      const mst_head_grads* grads, float* dpool_in, void* workspace, size_t workspace_bytes, void* stream);

/* The keep mask those Dropouts use, materialised: keep[i] = 1 iff element i (row-major index in the dropped tensor) survives
 * Dropout(p) under `seed`; survivors are scaled by 1 / (1 - p).  For tests and for callers that want the mask itself.   */
int mst_dropout_mask(float p, uint64_t seed, long long n, unsigned char* keep, void* stream);

/* FiLM MLP = MixingFeatureEncoder.forward (src/model.py:410-464) in training mode:
 * film = head(ReLU(mlp3(Dropout(ReLU(mlp0 feats)))));  feats: dev [B][feature_dim];  film: dev [B][out_dim] (n_sub * 192).
 * mst_film_backward: dfilm [B][out_dim] -> the six parameter gradients (overwritten); the features get no gradient.      */
typedef struct mst_film_dims { int32_t feature_dim, hidden, out_dim; } mst_film_dims;                /* feature_dim, hidden <= 2048 */
typedef struct mst_film_weights {
  const float *mlp0_w, *mlp0_b;   /* feature_mlp.0 [H][Fd], [H] */
  const float *mlp3_w, *mlp3_b;   /* feature_mlp.3 [H][H], [H]  */
  const float *head_w, *head_b;   /* film_head     [O][H], [O]  */
} mst_film_weights;
typedef struct mst_film_grads { float *mlp0_w, *mlp0_b, *mlp3_w, *mlp3_b, *head_w, *head_b; } mst_film_grads;
size_t mst_film_save_bytes(const mst_film_dims* dims, int B);
size_t mst_film_backward_workspace_bytes(const mst_film_dims* dims, int B);
int mst_film_forward_train(const mst_film_dims* dims, const mst_film_weights* w, const float* feats, int B, float drop_p,
                           uint64_t drop_seed, float* film, void* save, size_t save_bytes, void* stream);
int mst_film_backward(const mst_film_dims* dims, const mst_film_weights* w, const float* feats, int B, float drop_p,
                      const float* dfilm, const void* save, const mst_film_grads* grads, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * InfoNCE forward on (gathered) embeddings.  Replaces InfoNCELoss.forward src/loss.py:31-136.
 * emb: dev [N][D] fp32; labels: dev [N] int64; anchors [row0,row0+rows) are the local rows.
 * out: dev [2] = { sum over valid local anchors of -log(pos/(pos+neg+1e-8)), #valid anchors }.
 * ------------------------------------------------------------------------------------------ */
size_t mst_infonce_workspace_bytes(int N, int D);
int mst_infonce_forward(const float* emb, const int64_t* labels, int N, int D, int row0, int rows,
                        float temperature, float* out, void* workspace, size_t workspace_bytes,
                        void* stream);
/* Backward of the same sum (what autograd derives for src/loss.py:110-136):
 * grad: dev [N][D], overwritten with  (*scale) * d( sum over valid local anchors of loss_i ) / d emb  for ALL N rows
 *       (rows of other ranks receive the terms where they act as columns; the caller all-reduces and slices).
 * scale: dev [1] (the upstream gradient, e.g. 1/#valid anchors) or NULL for 1.  Same workspace size as forward. */
int mst_infonce_backward(const float* emb, const int64_t* labels, int N, int D, int row0, int rows,
                         float temperature, const float* scale, float* grad, void* workspace,
                         size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MST_H_ */
