"""MI355X mirror of reference src/model.py: MelSpectrogramPreprocessor, BandSplitEncoder, MixingFeatureEncoder,
MixingStyleEncoder (same constructor arguments, attributes and state_dict keys, so reference checkpoints load
with strict=True and reference call sites -- train.py:253,299,410, validation_utils.py:101 -- work unchanged).

Forward paths
  * inference (`model.eval()` or `torch.no_grad()`): hand-written HIP kernels through libmst.so --
    stage A (STFT -> mel -> log) and stage B (FiLM MLP, band-split conv stack, attention pooling).
    This is the product path and the one every parity claim refers to.
  * training with autograd (`model.train()` and grad enabled, `train_backend = "hip"`, the default): stage A in HIP (the
    waveform needs no gradient); the conv trunk -- train-mode BatchNorm forward, Dropout, and the whole backward (pool /
    ReLU / FiLM / BatchNorm, conv2 input gradient, both conv weight gradients) -- in hand-written HIP kernels behind one
    autograd Function (`_HipTrunk`); pooling head and FiLM MLP in csrc/head.hip.  A call the hand-written trunk cannot take
    RAISES (no silent library path); the library paths are explicit opt-ins: `train_backend = "torch"` runs the whole encoder on
    PyTorch-ROCm autograd (the path the gradients are tested against), `"hip-or-torch"` warns once per reason and falls back.
"""
import ctypes as C

import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .mixing_utils import (STEMS, LogMel, MelFeatPlan, detailed_bins_for_feature_dim, hann_window, is_deferred,
                           melscale_fbanks_htk, stems_to_tensor)


class _Buf(nn.Module):
    def __init__(self, name, value):
        super().__init__()
        self.register_buffer(name, value)


class _MelTransform(nn.Module):
    """Holds torchaudio's persistent buffers under their checkpoint names
    (`mel_transform.spectrogram.window`, `mel_transform.mel_scale.fb`)."""

    def __init__(self, sample_rate, n_fft, n_mels):
        super().__init__()
        self.spectrogram = _Buf("window", hann_window(n_fft))
        self.mel_scale = _Buf("fb", melscale_fbanks_htk(n_fft // 2 + 1, n_mels, sample_rate))


class MelSpectrogramPreprocessor(nn.Module):
    """Dict of 4 stems, each (B, 2, T) -> log-mel (B, 8, n_mels, frames).  reference src/model.py:17-67"""

    def __init__(self, sample_rate=44100, n_fft=1024, hop_length=256, n_mels=128):
        super().__init__()
        self.sample_rate, self.n_fft, self.hop_length, self.n_mels = sample_rate, n_fft, hop_length, n_mels
        self.mel_transform = _MelTransform(sample_rate, n_fft, n_mels)
        self._plans = {}

    def plan(self, detailed_bins=0) -> MelFeatPlan:
        """Stage-A plan; `detailed_bins` selects the feature layout the same launch can emit next to the log-mel."""
        if detailed_bins not in self._plans:  # built from the module's own buffers: a loaded checkpoint's tables are honoured
            self._plans[detailed_bins] = MelFeatPlan(self.sample_rate, self.n_fft, self.hop_length, self.n_mels,
                                                     detailed_bins, self.mel_transform.spectrogram.window,
                                                     self.mel_transform.mel_scale.fb)
        return self._plans[detailed_bins]

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._plans = {}

    def forward(self, stems_dict):
        lm, _ = self.plan().forward_stems(stems_dict, True, False)
        return lm


class FiLMLayer(nn.Module):
    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features

    def forward(self, x, gamma, beta):
        return gamma[:, :, None, None] * x + beta[:, :, None, None]


class SubSpectrogramCNN(nn.Module):
    """One band-split branch (reference src/model.py:97-157); parameters only -- the eval forward of all branches
    runs fused in the HIP encoder."""

    def __init__(self, split_size, channels, out_channels=64):
        super().__init__()
        self.split_size, self.channels, self.out_channels = split_size, channels, out_channels
        sub = max(1, split_size // 10)
        self.conv1 = nn.Conv2d(channels, 32, kernel_size=7, padding=3)
        self.bn1 = nn.BatchNorm2d(32)
        self.film1 = FiLMLayer(32)
        self.pool1 = nn.MaxPool2d((sub, 5))
        self.dropout1 = nn.Dropout(0.3)
        self.conv2 = nn.Conv2d(32, out_channels, kernel_size=7, padding=3)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.film2 = FiLMLayer(out_channels)
        self.pool2 = nn.MaxPool2d((4, 4))
        self.dropout2 = nn.Dropout(0.3)

    def forward(self, x, gamma1=None, beta1=None, gamma2=None, beta2=None):
        for conv, bn, film, pool, drop, g, b in ((self.conv1, self.bn1, self.film1, self.pool1, self.dropout1, gamma1,
                                                  beta1), (self.conv2, self.bn2, self.film2, self.pool2, self.dropout2,
                                                           gamma2, beta2)):
            x = bn(conv(x))
            if g is not None and b is not None:
                x = film(x, g, b)
            x = drop(pool(F.relu(x)))
        return x


class AttentionPooling(nn.Module):
    """reference src/model.py:160-211"""

    def __init__(self, input_dim, hidden_dim=128, output_dim=768):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.attention = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.Tanh(), nn.Linear(hidden_dim, 1))
        self.projection = nn.Sequential(nn.Linear(input_dim, output_dim), nn.ReLU(), nn.Dropout(0.3))

    def forward(self, x):
        xt = x.transpose(1, 2)
        w = F.softmax(self.attention(xt), dim=1)
        return self.projection((xt * w).sum(dim=1))


def count_subbands(n_mels, split_size, overlap):
    return len(range(0, n_mels - split_size + 1, overlap))


class BandSplitEncoder(nn.Module):
    """reference src/model.py:214-382"""

    def __init__(self, sample_rate=44100, n_fft=1024, hop_length=256, n_mels=128, split_size=20, overlap=10,
                 channels=8, embed_dim=768, cnn_out_channels=64):
        super().__init__()
        if channels != 8 or cnn_out_channels != 64:
            raise ValueError("the HIP encoder is built for 8 input channels (4 stems x stereo) and 64 CNN outputs")
        self.sample_rate, self.n_fft, self.hop_length, self.n_mels = sample_rate, n_fft, hop_length, n_mels
        self.split_size, self.overlap, self.channels, self.cnn_out_channels = split_size, overlap, channels, 64
        self.mel_preprocessor = MelSpectrogramPreprocessor(sample_rate, n_fft, hop_length, n_mels)
        self.n_subbands = count_subbands(n_mels, split_size, overlap)
        self.subnet_cnns = nn.ModuleList([SubSpectrogramCNN(split_size, channels, 64) for _ in range(self.n_subbands)])
        sub = max(1, split_size // 10)
        frames_10s = int(10.0 * sample_rate) // hop_length + 1
        self.freq_dim = (split_size // sub) // 4
        self.time_dim = (frames_10s // 5) // 4
        self.attention_pooling = AttentionPooling(64 * self.n_subbands * self.freq_dim, hidden_dim=256,
                                                  output_dim=embed_dim)

    def forward_from_logmel(self, x, film_params=None):
        """PyTorch-ROCm op path (autograd-capable).  x (B, 8, n_mels, frames)."""
        outs = []
        for i, cnn in enumerate(self.subnet_cnns):
            fp = film_params or {}
            outs.append(cnn(x[:, :, i * self.overlap:i * self.overlap + self.split_size, :], fp.get(f"gamma1_{i}"),
                            fp.get(f"beta1_{i}"), fp.get(f"gamma2_{i}"), fp.get(f"beta2_{i}")))
        y = torch.cat(outs, dim=1)
        return self.attention_pooling(y.reshape(y.shape[0], y.shape[1] * y.shape[2], y.shape[3]))

    def forward(self, stems_dict, film_params=None):
        return self.forward_from_logmel(self.mel_preprocessor(stems_dict), film_params)


class MixingFeatureEncoder(nn.Module):
    """reference src/model.py:385-464"""

    def __init__(self, feature_dim, n_subbands, hidden_dim=256):
        super().__init__()
        self.feature_dim, self.n_subbands = feature_dim, n_subbands
        self.feature_mlp = nn.Sequential(nn.Linear(feature_dim, hidden_dim), nn.ReLU(), nn.Dropout(0.2),
                                         nn.Linear(hidden_dim, hidden_dim), nn.ReLU())
        self.film_head = nn.Linear(hidden_dim, n_subbands * 192)

    def forward(self, features):
        flat = self.film_head(self.feature_mlp(features))
        out = {}
        for i in range(self.n_subbands):
            g1, b1, g2, b2 = torch.split(flat[:, i * 192:(i + 1) * 192], [32, 32, 64, 64], dim=1)
            out[f"gamma1_{i}"], out[f"beta1_{i}"], out[f"gamma2_{i}"], out[f"beta2_{i}"] = g1, b1, g2, b2
        return out


class HipEncoder:
    """Owns an `mst_encoder` handle built from a module's current parameters (eval-mode forward in HIP)."""

    def __init__(self, model: "MixingStyleEncoder", conv1_precision="fp32"):
        ae, fe = model.audio_encoder, model.film_encoder
        cpu = lambda t: t.detach().float().cpu().contiguous()
        cat = lambda name: cpu(torch.stack([getattr(getattr(c, name.split(".")[0]), name.split(".")[1])
                                            for c in ae.subnet_cnns], 0))
        self._keep = dict(
            conv1_w=cat("conv1.weight"), conv1_b=cat("conv1.bias"), bn1_w=cat("bn1.weight"), bn1_b=cat("bn1.bias"),
            bn1_mean=cat("bn1.running_mean"), bn1_var=cat("bn1.running_var"),
            conv2_w=cat("conv2.weight"), conv2_b=cat("conv2.bias"), bn2_w=cat("bn2.weight"), bn2_b=cat("bn2.bias"),
            bn2_mean=cat("bn2.running_mean"), bn2_var=cat("bn2.running_var"),
            mlp0_w=cpu(fe.feature_mlp[0].weight), mlp0_b=cpu(fe.feature_mlp[0].bias),
            mlp3_w=cpu(fe.feature_mlp[3].weight), mlp3_b=cpu(fe.feature_mlp[3].bias),
            head_w=cpu(fe.film_head.weight), head_b=cpu(fe.film_head.bias),
            att0_w=cpu(ae.attention_pooling.attention[0].weight), att0_b=cpu(ae.attention_pooling.attention[0].bias),
            att2_w=cpu(ae.attention_pooling.attention[2].weight), att2_b=cpu(ae.attention_pooling.attention[2].bias),
            proj_w=cpu(ae.attention_pooling.projection[0].weight), proj_b=cpu(ae.attention_pooling.projection[0].bias))
        w = _lib.EncoderWeights(**{k: v.data_ptr() for k, v in self._keep.items()})
        self.cfg = _lib.EncoderConfig(ae.n_mels, ae.split_size, ae.overlap, ae.n_subbands, fe.feature_dim,
                                      ae.attention_pooling.output_dim, fe.feature_mlp[0].out_features,
                                      ae.attention_pooling.attention[0].out_features, float(ae.subnet_cnns[0].bn1.eps))
        h = C.c_void_p()
        _lib.check(_lib.lib().mst_encoder_create(C.byref(h), C.byref(self.cfg), C.byref(w)), "mst_encoder_create")
        self._h = h
        modes = {"fp32": 0, "f16x3": 1, "f16x3-all": 2, "f16": 3}
        if conv1_precision not in modes:
            raise ValueError("conv precision must be 'fp32' (exact, default), 'f16x3' (conv1 on split-precision f16 MFMA), "
                             "'f16x3-all' (conv1 and conv2) or 'f16' (plain f16 operands, fp32 accumulate: the "
                             "arithmetic of the reference's --use_amp convolutions)")
        self.mode = modes[conv1_precision]
        _lib.check(_lib.lib().mst_encoder_set_precision(h, self.mode), "mst_encoder_set_precision")
        self._ws = None
        self.train_mode, self.train_f16 = 0, False   # precision of the training kernels (set_train_precision)
        self.embed_dim = ae.attention_pooling.output_dim
        self.n_sub, self.split, self.freq_dim, self.overlap = ae.n_subbands, ae.split_size, ae.freq_dim, ae.overlap
        self.sub = max(1, ae.split_size // 10)   # pool height of the first max-pool

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().mst_encoder_destroy(h)
            except Exception:
                pass

    def update_from(self, model: "MixingStyleEncoder"):
        """Refresh every table of this encoder from the module's CURRENT parameters and buffers ON THE DEVICE
        (`mst_encoder_update_params`): what `hip_encoder()` does before a validation pass that follows optimizer steps
        (src/train.py:388-427 after :292-296) -- no parameter crosses to the host, nothing synchronises.  The module must live
        on a GPU and have the shapes this encoder was created with."""
        ae, fe = model.audio_encoder, model.film_encoder
        dev = fe.film_head.weight.device
        if dev.type != "cuda":
            raise _lib.MstError("HipEncoder.update_from needs the module on a GPU")
        f32 = lambda t: t.detach().float().contiguous()
        cat = lambda name: f32(torch.stack([getattr(getattr(c, name.split(".")[0]), name.split(".")[1]).detach()
                                            for c in ae.subnet_cnns], 0))
        keep = dict(
            conv1_w=cat("conv1.weight"), conv1_b=cat("conv1.bias"), bn1_w=cat("bn1.weight"), bn1_b=cat("bn1.bias"),
            bn1_mean=cat("bn1.running_mean"), bn1_var=cat("bn1.running_var"),
            conv2_w=cat("conv2.weight"), conv2_b=cat("conv2.bias"), bn2_w=cat("bn2.weight"), bn2_b=cat("bn2.bias"),
            bn2_mean=cat("bn2.running_mean"), bn2_var=cat("bn2.running_var"),
            mlp0_w=f32(fe.feature_mlp[0].weight), mlp0_b=f32(fe.feature_mlp[0].bias),
            mlp3_w=f32(fe.feature_mlp[3].weight), mlp3_b=f32(fe.feature_mlp[3].bias),
            head_w=f32(fe.film_head.weight), head_b=f32(fe.film_head.bias),
            att0_w=f32(ae.attention_pooling.attention[0].weight), att0_b=f32(ae.attention_pooling.attention[0].bias),
            att2_w=f32(ae.attention_pooling.attention[2].weight), att2_b=f32(ae.attention_pooling.attention[2].bias),
            proj_w=f32(ae.attention_pooling.projection[0].weight), proj_b=f32(ae.attention_pooling.projection[0].bias))
        expect = (self.n_sub, 32, 8, 7, 7)
        if tuple(keep["conv1_w"].shape) != expect or keep["proj_w"].shape[0] != self.embed_dim:
            raise _lib.MstError(f"HipEncoder.update_from: parameter shapes {tuple(keep['conv1_w'].shape)} differ from the encoder's {expect}")
        w = _lib.EncoderWeights(**{k: v.data_ptr() for k, v in keep.items()})
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mst_encoder_update_params(self._h, C.byref(w), _lib.stream_ptr(dev)), "mst_encoder_update_params")
        # (the stacked temporaries are released to the caching allocator here; the kernels that read them are already queued on
        # this stream, and the allocator re-issues a block only to work queued later on the same stream)

    def _ws_view(self, offset, nbytes, dtype):
        return self._ws_train[offset:offset + nbytes].view(dtype)

    def stats_view(self, layer, B, frames):
        """int64 view of the statistics accumulators of conv layer 1 / 2 inside the training workspace (forward: sum y, sum y^2;
        backward: sum dz, sum dz * zhat) -- what data-parallel ranks add up for cross-rank BatchNorm statistics."""
        off, n = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.lib().mst_encoder_train_stats_buffer(self._h, layer, B, frames, C.byref(off), C.byref(n)),
                   "mst_encoder_train_stats_buffer")
        return self._ws_view(off.value, 8 * n.value, torch.int64)

    def scale_view(self, B, frames):
        """int32 view of the word that holds max |d pool_in| (float bits, non-negative) in the f16 training modes."""
        off = C.c_size_t()
        _lib.check(_lib.lib().mst_encoder_train_scale_buffer(self._h, B, frames, C.byref(off)), "mst_encoder_train_scale_buffer")
        return self._ws_view(off.value, 4, torch.int32)

    def forward_train(self, logmel, feats=None, film=None, head=True, drop1_mask=None, drop1_p=0.0, sync=None, drop1_seed=None,
                      want_pool1=True, want_film=True, want_bn=True):
        """`forward_train_steps` run to completion; `sync` (an object with `.world`, `.sum(int64 tensor)`, `.max(int32 tensor)`,
        e.g. DistSync) adds the ranks' BatchNorm statistics between the phases; None = this process only."""
        steps = self.forward_train_steps(logmel, feats, film, head, drop1_mask, drop1_p, sync.world if sync is not None else 0,
                                         drop1_seed, want_pool1, want_film, want_bn)
        try:
            while True:
                kind, view = next(steps)
                getattr(sync, kind)(view)
        except StopIteration as done:
            return done.value

    @staticmethod
    def _logmel_in(logmel):
        """(mst_logmel_in, tensors to keep alive, (B, frames), device) of a reference-layout tensor or a `LogMel`."""
        if isinstance(logmel, LogMel):
            return (_lib.LogmelIn(logmel.layout, 0, _lib.dptr(logmel.data), _lib.dptr(logmel.lo), _lib.dptr(logmel.absmax)), logmel,
                    (logmel.B, logmel.frames), logmel.device)
        lm = logmel.contiguous().float()
        return _lib.LogmelIn(_lib.LOGMEL_REF, 0, _lib.dptr(lm), None, None), lm, (lm.shape[0], lm.shape[3]), lm.device

    def train_layout(self):
        """The log-mel layout the TRAINING kernels of the current precision read fastest: stage A's float16 planes
        (_lib.LOGMEL_CM16) in the f16 / f16x3 modes, the reference layout in the fp32 mode."""
        lay = _lib.LOGMEL_CM16
        return lay if _lib.lib().mst_encoder_train_layout_supported(self._h, lay) else _lib.LOGMEL_REF

    def forward_train_steps(self, logmel, feats=None, film=None, head=True, drop1_mask=None, drop1_p=0.0, world=0,
                            drop1_seed=None, want_pool1=True, want_film=True, want_bn=True):
        """Generator form of the train-mode forward.  world = 0: one call of `mst_encoder_forward_train`, nothing is yielded.
        world >= 1: the three phases of include/mst.h; after phase 1 and 2 it yields ("sum", int64 view of that layer's
        statistics accumulators), which the caller must all-reduce (SUM) over its `world` ranks before resuming.
        Returns (via StopIteration.value) what `forward_train` returns.
        Train-mode forward in libmst.so (`mst_encoder_forward_train`): BatchNorm with batch statistics.
        feats (B, Fd): FiLM MLP in HIP; or film (B, n_sub*192): FiLM parameters from the caller's own MLP.
        head=False stops at pool_in (emb is None).  drop1_mask: uint8 keep-mask shaped like pool1 (Dropout after the
        first pooling); or drop1_seed (int) with drop1_p > 0: the kernel draws the mask itself (Philox, a pure function of
        (seed, element)) and returns it as taps["drop1_mask"].  Returns (emb, taps) with taps = film, pool1, pool_in, bn1, bn2 ((n_sub, C, 2): batch mean and
        1/sqrt(biased var + eps)).  The raw conv outputs stay in the workspace for `backward_apply`.
        want_pool1=False (float16 training modes): the fp32 pool1 is not written -- conv2 and its weight gradient read the float16
        planes the pooling epilogue leaves in the workspace (`conv2_wgrad(None, ...)`)."""
        lin, keep, (B, Fr), dev = self._logmel_in(logmel)
        L = _lib.lib()
        need = L.mst_encoder_train_workspace_bytes(self._h, B, Fr)
        if need == 0:
            raise _lib.MstError("forward_train needs frames >= 20")
        if getattr(self, "_ws_train", None) is None or self._ws_train.numel() < need or self._ws_train.device != dev:
            self._ws_train = torch.empty(need, dtype=torch.uint8, device=dev)
        if _POISON_WS:   # debugging aid: every byte 0xFF (fp32 NaN) before each forward -- any read of a slot nobody wrote shows up
            self._ws_train.fill_(255)
        emb = torch.empty(B, self.embed_dim, dtype=torch.float32, device=dev) if head else None
        W1 = Fr // 5
        p1_shape = (B, self.n_sub, 32, self.split // max(1, self.split // 10), W1)
        if not want_pool1 and not self.train_f16:
            raise _lib.MstError("forward_train(want_pool1=False) needs a float16 training mode (the fp32 conv2 reads the fp32 pool1)")
        out = {"film": torch.empty(B, self.n_sub * 192, device=dev) if want_film else None,
               "pool1": torch.empty(p1_shape, device=dev) if want_pool1 else None,
               "pool_in": torch.empty(B, 64 * self.n_sub * self.freq_dim, W1 // 4, device=dev),
               "bn1": torch.empty(self.n_sub, 32, 2, device=dev) if want_bn else None,
               "bn2": torch.empty(self.n_sub, 64, 2, device=dev) if want_bn else None}
        film_c = film.detach().contiguous().float() if film is not None else None
        feats_c = feats.contiguous().float() if feats is not None else None
        mask_c = drop1_mask.contiguous() if drop1_mask is not None else None
        if mask_c is not None:
            assert mask_c.dtype == torch.uint8 and tuple(mask_c.shape) == p1_shape
        mask_out = None
        if drop1_seed is not None and drop1_p > 0.0:
            assert mask_c is None, "pass a mask OR a seed"
            mask_out = out["drop1_mask"] = torch.empty(p1_shape, dtype=torch.uint8, device=dev)
        for phase in ((0,) if world == 0 else (1, 2, 3)):
            t = _lib.EncoderTrainTaps(*[_lib.dptr(out[k]) for k in ("film", "pool1", "pool_in", "bn1", "bn2")],
                                      _lib.dptr(film_c), _lib.dptr(mask_c), 1.0 / (1.0 - drop1_p) if mask_c is not None else 1.0,
                                      phase, float(max(world, 1)), _lib.dptr(mask_out), int(drop1_seed or 0) & (2 ** 64 - 1),
                                      float(drop1_p) if mask_out is not None else 0.0)
            with torch.cuda.device(dev):
                _lib.check(L.mst_encoder_forward_train_in(self._h, C.byref(lin), Fr, _lib.dptr(feats_c), B, _lib.dptr(emb),
                                                          C.byref(t), _lib.dptr(self._ws_train), need, _lib.stream_ptr(dev)),
                           "mst_encoder_forward_train_in")
            if phase in (1, 2):
                yield "sum", self.stats_view(phase, B, Fr)
        return emb, out

    def _zero_bias_grad(self, c, dev):
        """(n_sub, c) zeros: the gradient of a convolution bias in front of a batch-statistics BatchNorm.  A fresh tensor every time:
        it becomes a `.grad` the caller owns and may modify in place."""
        return torch.zeros(self.n_sub, c, device=dev)

    def update_running_stats(self, layer, B, frames, running_mean, running_var, num_batches_tracked, momentum, cross_rank=False):
        """nn.BatchNorm2d's running-statistics update of layer 1 / 2 from the batch statistics the last `forward_train` left in its
        workspace (`mst_encoder_train_update_running_stats`: one launch; stacked buffers (n_sub, C) / (n_sub,) int64)."""
        L = _lib.lib()
        need = L.mst_encoder_train_workspace_bytes(self._h, B, frames)
        dev = running_mean.device
        assert running_mean.is_contiguous() and running_var.is_contiguous() and num_batches_tracked.dtype == torch.int64
        with torch.cuda.device(dev):
            _lib.check(L.mst_encoder_train_update_running_stats(self._h, layer, B, frames, _lib.dptr(running_mean), _lib.dptr(running_var),
                                                                _lib.dptr(num_batches_tracked), float(momentum), int(bool(cross_rank)),
                                                                _lib.dptr(self._ws_train), need, _lib.stream_ptr(dev)),
                       "mst_encoder_train_update_running_stats")

    TRAIN_MODES = {"fp32": 0, "f16": 1, "f16x3": 2}

    def set_train_precision(self, mode):
        """Precision of the training kernels (`mst_encoder_set_train_precision`): "fp32" / False = exact fp32 MFMA;
        "f16" / True = float16 operands with fp32 accumulation in the forward and backward convolutions (the reference's
        --use_amp arithmetic); "f16x3" = the same kernels with 3-term split-precision operands (hi + lo float16 pairs,
        fp32-equivalent results at ~5x the fp32 matrix rate).  Takes effect with the next `update_trunk_params` +
        `forward_train` pair."""
        m = self.TRAIN_MODES[mode] if isinstance(mode, str) else (1 if mode else 0)
        if m != self.train_mode:
            _lib.check(_lib.lib().mst_encoder_set_train_precision(self._h, m), "mst_encoder_set_train_precision")
            self.train_mode = m
            self.train_f16 = m != 0   # the backward runs on the f16 kernels (f16 dy buffers; mode 2: hi and lo)

    def update_trunk_params(self, c1w, c1b, bn1w, bn1b, c2w, c2b, bn2w, bn2b):
        """Refresh the training kernels' conv / BatchNorm parameters from device tensors stacked over the sub-bands."""
        ts = [t.detach().contiguous().float() for t in (c1w, c1b, bn1w, bn1b, c2w, c2b, bn2w, bn2b)]
        with torch.cuda.device(ts[0].device):
            _lib.check(_lib.lib().mst_encoder_update_trunk_params(self._h, *[_lib.dptr(t) for t in ts],
                                                                  _lib.stream_ptr(ts[0].device)),
                       "mst_encoder_update_trunk_params")

    def conv1_wgrad(self, logmel, B, frames):
        """conv1 weight gradient (n_sub, 32, 8, 7, 7) from the d(conv1 output) that `backward_apply(1, ..., inplace=True)`
        left in the workspace (`mst_encoder_train_conv1_wgrad`)."""
        L = _lib.lib()
        lin, keep, _, dev = self._logmel_in(logmel)   # the SAME log-mel (layout included) the forward pass read
        dw = torch.empty(self.n_sub, 32, 8, 7, 7, device=dev)
        need = L.mst_encoder_train_workspace_bytes(self._h, B, frames)
        with torch.cuda.device(dev):
            _lib.check(L.mst_encoder_train_conv1_wgrad_in(self._h, C.byref(lin), B, frames, _lib.dptr(dw),
                                                          _lib.dptr(self._ws_train), need, _lib.stream_ptr(dev)),
                       "mst_encoder_train_conv1_wgrad_in")
        return dw

    def conv2_dgrad(self, dy2, B, frames, mask=None, drop_p=0.0):
        """gradient of pool1 (B, n_sub, 32, H1, W1) from dy2 (n_sub, B, 64, H1, W1) -- f16 training: the float16
        (n_sub, B, H1, W1, 64) tensor `backward_apply(2, ...)` returned: `mst_encoder_train_conv2_dgrad`."""
        assert dy2.dtype == (torch.float16 if getattr(self, "train_f16", False) else torch.float32)
        L = _lib.lib()
        out = torch.empty(B, self.n_sub, 32, self.split // self.sub, frames // 5, device=dy2.device)
        with torch.cuda.device(dy2.device):
            _lib.check(L.mst_encoder_train_conv2_dgrad(self._h, _lib.dptr(dy2), B, frames, _lib.dptr(out), _lib.dptr(mask),
                                                       1.0 / (1.0 - drop_p) if mask is not None else 1.0,
                                                       _lib.stream_ptr(dy2.device)), "mst_encoder_train_conv2_dgrad")
        return out

    def conv2_wgrad(self, pool1, B, frames):
        """conv2 weight gradient (n_sub, 64, 32, 7, 7) from the accumulator-order d(conv2 output) in the workspace.
        pool1 None (float16 training modes): the operand is the float16 pool1 the forward pass left in the workspace."""
        L = _lib.lib()
        dev = self._ws_train.device
        dw = torch.empty(self.n_sub, 64, 32, 7, 7, device=dev)
        need = L.mst_encoder_train_workspace_bytes(self._h, B, frames)
        p1 = pool1.contiguous().float() if pool1 is not None else None
        with torch.cuda.device(dev):
            _lib.check(L.mst_encoder_train_conv2_wgrad(self._h, _lib.dptr(p1), B, frames, _lib.dptr(dw),
                                                       _lib.dptr(self._ws_train), need, _lib.stream_ptr(dev)),
                       "mst_encoder_train_conv2_wgrad")
        return dw

    def backward_apply(self, layer, dpool, dfilm, B, frames, inplace=False, sync=None):
        """`backward_apply_steps` run to completion (see `forward_train` for `sync`)."""
        steps = self.backward_apply_steps(layer, dpool, dfilm, B, frames, inplace, sync.world if sync is not None else 0)
        try:
            while True:
                kind, view = next(steps)
                getattr(sync, kind)(view)
        except StopIteration as done:
            return done.value

    def backward_apply_steps(self, layer, dpool, dfilm, B, frames, inplace=False, world=0):
        """Generator form: world = 0 -> one call, nothing yielded; world >= 1 -> phases 1..3 of
        `mst_encoder_train_backward_apply_phase`, yielding ("max", int32 view) after phase 1 (f16 training modes, layer 2: the
        ranks must agree on the internal loss scale) and ("sum", int64 view of the layer's sums) after phase 2.
        The returned dbn are THIS rank's contributions (the trainer's gradient all-reduce adds the ranks up).
        Backward of pool/ReLU/FiLM/BatchNorm(train) of conv layer 1 or 2 from the activations the last
        `forward_train` call left in its workspace (`mst_encoder_train_backward_apply`).
        dpool: layer 1 (B, n_sub, 32, 10, W1) or any tensor with those dims and arbitrary clip / band / channel
        strides; layer 2 (B, 64*n_sub*freq_dim, W2) = d pool_in.  dfilm (B, n_sub*192) is accumulated in place.
        Returns (dy (n_sub, B, C, rows, cols), dbn (n_sub, C, 2) = (d weight, d bias))."""
        L = _lib.lib()
        dev = dpool.device
        W1 = frames // 5
        if layer == 1:
            assert dpool.dim() == 5 and dpool.stride(4) == 1 and dpool.stride(3) == W1
            st = (dpool.stride(0), dpool.stride(1), dpool.stride(2))
            dy = None if inplace else torch.empty(self.n_sub, B, 32, self.split, frames, device=dev)
            dbn = torch.empty(2, self.n_sub, 32, device=dev)   # planes: d weight | d bias
        else:
            dpool = dpool.contiguous()
            W2 = W1 // 4
            st = (dpool.shape[1] * W2, 64 * self.freq_dim * W2, self.freq_dim * W2)
            if getattr(self, "train_f16", False):   # the f16 dgrad kernel's operand: channel-minor halves (f16x3: hi, then lo)
                dy = torch.empty((2,) * (self.train_mode == 2) + (self.n_sub, B, self.split // self.sub, W1, 64), device=dev,
                                 dtype=torch.float16)
            else:
                dy = torch.empty(self.n_sub, B, 64, self.split // self.sub, W1, device=dev)
            dbn = torch.empty(2, self.n_sub, 64, device=dev)
        need = L.mst_encoder_train_workspace_bytes(self._h, B, frames)
        for phase in ((0,) if world == 0 else (1, 2, 3)):
            with torch.cuda.device(dev):
                _lib.check(L.mst_encoder_train_backward_apply_phase(self._h, layer, B, frames, _lib.dptr(dpool), st[0], st[1], st[2],
                                                                    _lib.dptr(dy), _lib.dptr(dfilm), _lib.dptr(dbn),
                                                                    _lib.dptr(self._ws_train), need, _lib.stream_ptr(dev), phase,
                                                                    float(max(world, 1))),
                           "mst_encoder_train_backward_apply")
            if phase == 1 and layer == 2 and self.train_f16:
                yield "max", self.scale_view(B, frames)
            elif phase == 2:
                yield "sum", self.stats_view(layer, B, frames)
        return dy, dbn.permute(1, 2, 0)   # (n_sub, C, 2) view: [..., 0] / [..., 1] are the contiguous stacked gradients

    def preferred_layout(self):
        """The log-mel layout this encoder's eval forward reads fastest (_lib.LOGMEL_*): the channel-minor form of its conv1
        precision when the kernel geometry takes it, else the reference layout."""
        L = _lib.lib()
        for lay in ((_lib.LOGMEL_CM16,) if self.mode else (_lib.LOGMEL_CM32,)):
            if L.mst_encoder_layout_supported(self._h, lay):
                return lay
        return _lib.LOGMEL_REF

    def forward(self, logmel, feats, taps=False, events=None):
        """logmel: (B, 8, n_mels, frames) tensor (reference layout) or a `LogMel` (channel-minor, from stage A).
        events: optional list of 6 recorded torch.cuda.Event (stage boundaries, see include/mst.h)."""
        if isinstance(logmel, LogMel):
            lin = _lib.LogmelIn(logmel.layout, 0, _lib.dptr(logmel.data), _lib.dptr(logmel.lo), _lib.dptr(logmel.absmax))
            B, Fr = logmel.B, logmel.frames
            keep = logmel   # noqa: F841  (the tensors stay alive until the launch is queued)
            logmel = logmel.data
        else:
            logmel = logmel.contiguous().float()
            lin = _lib.LogmelIn(_lib.LOGMEL_REF, 0, _lib.dptr(logmel), None, None)
            B, _, M, Fr = logmel.shape
        L = _lib.lib()
        need = L.mst_encoder_workspace_bytes(self._h, B, Fr)
        if self._ws is None or self._ws.numel() < need or self._ws.device != logmel.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=logmel.device)
        if _POISON_WS:
            self._ws.fill_(255)
        emb = torch.empty(B, self.embed_dim, dtype=torch.float32, device=logmel.device)
        t, out = None, {}
        if taps:
            sub = max(1, self.split // 10)
            H1, W1 = self.split // sub, Fr // 5
            out = dict(film=torch.empty(B, self.n_sub * 192, device=logmel.device),
                       pool1=torch.empty(B, self.n_sub, 32, H1, W1, device=logmel.device),
                       pool_in=torch.empty(B, 64 * self.n_sub * self.freq_dim, W1 // 4, device=logmel.device))
            t = C.byref(_lib.EncoderTaps(out["film"].data_ptr(), out["pool1"].data_ptr(), out["pool_in"].data_ptr()))
        elif events is not None:
            tp = _lib.EncoderTaps()
            for i, ev in enumerate(events):
                tp.events[i] = ev.cuda_event
            t = C.byref(tp)
        with torch.cuda.device(logmel.device):
            _lib.check(L.mst_encoder_forward_in(self._h, C.byref(lin), Fr, _lib.dptr(feats.contiguous().float()), B,
                                                _lib.dptr(emb), t, _lib.dptr(self._ws), need,
                                                _lib.stream_ptr(logmel.device)), "mst_encoder_forward_in")
        return (emb, out) if taps else emb


class DistSync:
    """Cross-rank sums for BatchNorm statistics over a torch.distributed group (RCCL): exact, order-independent integer adds of
    the kernels' accumulators.  Any object with `.world`, `.sum(int64 tensor)` and `.max(int32 tensor)` will do."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist, self.group, self.world = dist, group, dist.get_world_size(group)

    def sum(self, t):
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)

    def max(self, t):
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)


_TRAIN_TIMING = bool(os.environ.get("MST_TRAIN_TIMING"))
_POISON_WS = bool(os.environ.get("MST_POISON_WS"))


class _HipTrunk(torch.autograd.Function):
    """The 11 x [conv -> BatchNorm(batch statistics) -> FiLM -> ReLU -> max-pool] x 2 trunk for training.
    Forward: libmst.so (`mst_encoder_forward_train`, raw conv outputs kept).  Backward: pool / ReLU / FiLM / BatchNorm in
    libmst.so (`mst_encoder_train_backward_apply`), conv2 input gradient (`mst_encoder_train_conv2_dgrad`) and both weight
    gradients (`mst_encoder_train_conv{1,2}_wgrad`) as hand-written MFMA kernels.  (The library yardstick for these
    gradients is `train_backend = "torch"`: the whole encoder on PyTorch-ROCm autograd.)"""
    last_timing = None

    @staticmethod
    def forward(ctx, enc, logmel, film, flat, drop_p, sync, reducer, want_bn, *params):
        """flat: the 8 parameter families (conv1.weight, conv1.bias, bn1.weight, bn1.bias, conv2.*, bn2.*) stacked over the
        sub-bands -- the storage the per-band Parameters are views of (MixingStyleEncoder._trunk_flat), so no stack kernels run;
        params: those Parameters themselves, family-major, only so that autograd routes the gradients to them (the backward
        returns views of the stacked gradients, which AccumulateGrad adopts without a copy when .grad is None)."""
        if isinstance(logmel, LogMel):
            B, Fr = logmel.B, logmel.frames
        else:
            B, _, M, Fr = logmel.shape
        # this pass's parameter snapshot (an optimizer step or another pass may rewrite the live storage before this pass's backward
        # runs): ONE copy when the eight families live in one storage (MixingStyleEncoder._trunk_flat), else one per family
        base = getattr(flat[0], "_base", None)
        if base is not None and all(getattr(b, "_base", None) is base for b in flat):
            snap = base.clone()
            off = [b.storage_offset() - base.storage_offset() for b in flat]
            trunk = tuple(snap[o:o + b.numel()].view(b.shape) for o, b in zip(off, flat))
        else:
            trunk = tuple(b.clone() for b in flat)
        ctx.n_sub = flat[0].shape[0]
        enc.update_trunk_params(*trunk)
        # which forward pass's parameters the encoder's weight fragments hold: a MONOTONIC pass counter names the passes, the
        # owner is rewritten on every fragment rebuild (forward or backward) and never fed back into the counter
        enc._gen_counter = getattr(enc, "_gen_counter", 0) + 1
        enc._frag_owner = enc._gen_counter
        W1 = Fr // 5
        # Dropout after the first pooling (src/model.py:118): the mask is drawn inside the pooling epilogue (Philox keyed by a seed
        # taken from torch's CPU generator: reproducible under torch.manual_seed, no 87 M-element rand / compare / cast launches)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if drop_p > 0.0 else None
        # every forward pass gets its OWN activation workspace (the caching allocator hands the previous step's block back, so a
        # plain training loop allocates nothing): several forward passes may be alive at once -- retained graphs, two
        # forwards before one backward -- each backward finds its activations in the workspace its context holds
        enc._ws_train = None
        # (float16 modes: conv2 and its weight gradient read the float16 pool1 planes in the workspace -- no fp32 pool1 tensor)
        _, t = enc.forward_train(logmel, film=film, head=False, drop1_p=drop_p, sync=sync, drop1_seed=seed,
                                 want_pool1=not enc.train_f16, want_film=False, want_bn=want_bn)
        mask = t.get("drop1_mask")
        ctx.enc, ctx.drop_p, ctx.dims, ctx.sync, ctx.reducer = enc, drop_p, (B, Fr), sync, reducer
        ctx.trunk_params = params   # family-major (MixingStyleEncoder._TRUNK_FAMILIES): the reducer maps stacked gradients to their owners
        ctx.gen, ctx.mode = enc._gen_counter, enc.train_mode
        # (the workspace rides with the saved tensors: autograd releases it with them after a backward that does not retain the graph)
        if isinstance(logmel, LogMel):   # stage A's float16 planes: kept for conv1's weight gradient as they are
            ctx.lm_layout = logmel.layout
            ctx.save_for_backward(logmel.data, logmel.lo, t["pool1"], mask, enc._ws_train, *trunk)
        else:
            ctx.lm_layout = None
            ctx.save_for_backward(logmel, None, t["pool1"], mask, enc._ws_train, *trunk)
        if not want_bn:   # (the caller updates the running statistics from the workspace: mst_encoder_train_update_running_stats)
            return t["pool_in"]
        ctx.mark_non_differentiable(t["bn1"], t["bn2"])
        return t["pool_in"], t["bn1"], t["bn2"]

    @staticmethod
    def backward(ctx, dpool_in, _d1=None, _d2=None):
        enc, (B, Fr) = ctx.enc, ctx.dims
        logmel, lm_lo, p1, mask, ws, *trunk = ctx.saved_tensors
        dev = logmel.device
        if ctx.lm_layout is not None:
            logmel = LogMel(ctx.lm_layout, logmel, lm_lo)
        if getattr(ctx, "consumed", False):
            raise RuntimeError("HIP training trunk (fp32 mode): second backward() through the same forward pass -- the fp32 kernels "
                               "turn the saved convolution outputs into their gradients IN PLACE; run the forward again, or use "
                               "train_precision='f16x3' / 'f16' (their backward leaves the activations intact)")
        ctx.consumed = ctx.mode == 0
        enc._ws_train = ws                           # this pass's activations
        if ctx.gen != getattr(enc, "_frag_owner", None) or ctx.mode != enc.train_mode:   # another pass re-swizzled the fragments since:
            enc.set_train_precision({v: k for k, v in enc.TRAIN_MODES.items()}[ctx.mode])   # put this pass's parameters back
            enc.update_trunk_params(*trunk)
            enc._frag_owner = ctx.gen
        ns = enc.n_sub
        marks = []

        def mark(name):   # MST_TRAIN_TIMING=1: per-section GPU times of the last backward in _HipTrunk.last_timing
            if _TRAIN_TIMING:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((name, e))
        mark("start")
        dfilm = torch.zeros(B, ns * 192, device=dev)
        sync = ctx.sync
        dy2, dbn2 = enc.backward_apply(2, dpool_in.contiguous(), dfilm, B, Fr, sync=sync)
        mark("apply_bwd2")
        gw2 = enc.conv2_wgrad(p1, B, Fr)          # weight gradient on dy2 in accumulator order
        gb2 = enc._zero_bias_grad(64, dev)      # exactly 0 in front of a batch-statistics BatchNorm
        mark("conv2_wgrad")
        dbn2w, dbn2b = dbn2[..., 0], dbn2[..., 1]   # contiguous planes (no copies)
        reducer = ctx.reducer
        if reducer is not None:   # data parallel: the conv2-side gradients are final -- their all-reduce runs behind the rest of
            reducer.reduce_stacked("conv2", [gw2, gb2, dbn2w, dbn2b], [ctx.trunk_params[f * ns:(f + 1) * ns] for f in (4, 5, 6, 7)])   # this backward (dist.GradientReducer)
        dp1 = enc.conv2_dgrad(dy2, B, Fr, mask, ctx.drop_p)   # input gradient (Dropout mask fused)
        mark("conv2_dgrad")
        _, dbn1 = enc.backward_apply(1, dp1, dfilm, B, Fr, inplace=True, sync=sync)
        mark("apply_bwd1")
        gw1 = enc.conv1_wgrad(logmel, B, Fr)
        gb1 = enc._zero_bias_grad(32, dev)
        mark("conv1_wgrad")
        if _TRAIN_TIMING:
            torch.cuda.synchronize()
            _HipTrunk.last_timing = {b[0]: round(a[1].elapsed_time(b[1]), 3) for a, b in zip(marks[:-1], marks[1:])}
        dbn1w, dbn1b = dbn1[..., 0], dbn1[..., 1]
        if reducer is not None:
            reducer.reduce_stacked("conv1", [gw1, gb1, dbn1w, dbn1b], [ctx.trunk_params[f * ns:(f + 1) * ns] for f in (0, 1, 2, 3)])
        fams = (gw1, gb1, dbn1w, dbn1b, gw2, gb2, dbn2w, dbn2b)
        return (None, None, dfilm, None, None, None, None, None) + tuple(g[i] for g in fams for i in range(ns))


def _seed64():
    """A 62-bit seed from torch's CPU generator (reproducible under torch.manual_seed; no device work)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class _HipHead(torch.autograd.Function):
    """Dropout -> AttentionPooling (src/model.py:118, :187-211) for training, forward and backward in libmst.so
    (`mst_head_forward_train` / `mst_head_backward`, csrc/head.hip): fp32, deterministic, Dropout masks re-derived from
    (seed, element) instead of stored.  Parameters: attention.0 / attention.2 / projection.0 weight and bias."""

    @staticmethod
    def forward(ctx, pool_in, p_in, p_out, a0w, a0b, a2w, a2b, pw, pb):
        B, Cc, T = pool_in.shape
        x = pool_in.contiguous().float()
        ws = [t.detach().contiguous().float() for t in (a0w, a0b, a2w, a2b, pw, pb)]
        dims = _lib.HeadDims(Cc, T, ws[0].shape[0], ws[4].shape[0])
        L = _lib.lib()
        seeds = (_seed64() if p_in > 0.0 else 0, _seed64() if p_out > 0.0 else 0)
        save = torch.empty(L.mst_head_save_bytes(C.byref(dims), B, float(p_in)), dtype=torch.uint8, device=x.device)
        emb = torch.empty(B, ws[4].shape[0], device=x.device)
        wp = _lib.HeadPtrs(*[t.data_ptr() for t in ws])
        with torch.cuda.device(x.device):
            _lib.check(L.mst_head_forward_train(C.byref(dims), C.byref(wp), _lib.dptr(x), B, float(p_in), seeds[0], float(p_out), seeds[1],
                                                _lib.dptr(emb), _lib.dptr(save), save.numel(), _lib.stream_ptr(x.device)),
                       "mst_head_forward_train")
        ctx.cfg = (dims, float(p_in), float(p_out), seeds)
        ctx.save_for_backward(x, save, *ws)   # (the weight snapshots are the live tensors unless a dtype / layout copy was needed)
        return emb

    @staticmethod
    def backward(ctx, demb):
        x, save, *ws = ctx.saved_tensors
        dims, p_in, p_out, seeds = ctx.cfg
        B = x.shape[0]
        L = _lib.lib()
        grads = [torch.empty_like(t) for t in ws]
        dx = torch.empty_like(x)
        work = torch.empty(L.mst_head_backward_workspace_bytes(C.byref(dims), B), dtype=torch.uint8, device=x.device)
        de = demb.contiguous().float()
        wp, gp = _lib.HeadPtrs(*[t.data_ptr() for t in ws]), _lib.HeadPtrs(*[t.data_ptr() for t in grads])
        with torch.cuda.device(x.device):
            _lib.check(L.mst_head_backward(C.byref(dims), C.byref(wp), _lib.dptr(x), B, p_in, seeds[0], p_out, seeds[1], _lib.dptr(de),
                                           _lib.dptr(save), C.byref(gp), _lib.dptr(dx), _lib.dptr(work), work.numel(),
                                           _lib.stream_ptr(x.device)), "mst_head_backward")
        return (dx, None, None) + tuple(grads)


class _HipFilmMLP(torch.autograd.Function):
    """MixingFeatureEncoder's MLP + film_head (src/model.py:410-464) for training, forward and backward in libmst.so
    (`mst_film_forward_train` / `mst_film_backward`).  The features get no gradient (they are data)."""

    @staticmethod
    def forward(ctx, feats, p, w0, b0, w3, b3, wh, bh):
        f = feats.contiguous().float()
        B = f.shape[0]
        ws = [t.detach().contiguous().float() for t in (w0, b0, w3, b3, wh, bh)]
        dims = _lib.FilmDims(ws[0].shape[1], ws[0].shape[0], ws[4].shape[0])
        L = _lib.lib()
        seed = _seed64() if p > 0.0 else 0
        save = torch.empty(L.mst_film_save_bytes(C.byref(dims), B), dtype=torch.uint8, device=f.device)
        film = torch.empty(B, ws[4].shape[0], device=f.device)
        wp = _lib.FilmPtrs(*[t.data_ptr() for t in ws])
        with torch.cuda.device(f.device):
            _lib.check(L.mst_film_forward_train(C.byref(dims), C.byref(wp), _lib.dptr(f), B, float(p), seed, _lib.dptr(film), _lib.dptr(save),
                                                save.numel(), _lib.stream_ptr(f.device)), "mst_film_forward_train")
        ctx.cfg = (dims, float(p))
        ctx.save_for_backward(f, save, *ws)
        return film

    @staticmethod
    def backward(ctx, dfilm):
        f, save, *ws = ctx.saved_tensors
        dims, p = ctx.cfg
        B = f.shape[0]
        L = _lib.lib()
        grads = [torch.empty_like(t) for t in ws]
        work = torch.empty(L.mst_film_backward_workspace_bytes(C.byref(dims), B), dtype=torch.uint8, device=f.device)
        df = dfilm.contiguous().float()
        wp, gp = _lib.FilmPtrs(*[t.data_ptr() for t in ws]), _lib.FilmPtrs(*[t.data_ptr() for t in grads])
        with torch.cuda.device(f.device):
            _lib.check(L.mst_film_backward(C.byref(dims), C.byref(wp), _lib.dptr(f), B, p, _lib.dptr(df), _lib.dptr(save), C.byref(gp),
                                           _lib.dptr(work), work.numel(), _lib.stream_ptr(f.device)), "mst_film_backward")
        return (None, None) + tuple(grads)


class MixingStyleEncoder(nn.Module):
    """reference src/model.py:467-542.  `encoder_backend`: "hip" (default; eval/no-grad forward in libmst.so) or
    "torch" (PyTorch-ROCm ops for stage B; stage A stays HIP) -- BASELINE.json configs[2] vs configs[1]."""

    def __init__(self, sample_rate=44100, n_fft=1024, hop_length=256, n_mels=128, split_size=20, overlap=10,
                 channels=8, embed_dim=768, feature_dim=256, encoder_backend="hip"):
        super().__init__()
        self.audio_encoder = BandSplitEncoder(sample_rate, n_fft, hop_length, n_mels, split_size, overlap, channels,
                                              embed_dim)
        self.film_encoder = MixingFeatureEncoder(feature_dim, self.audio_encoder.n_subbands)
        self.encoder_backend = encoder_backend
        self.conv1_precision = "fp32"   # "f16x3": opt-in split-precision f16 MFMA for conv1 (see include/mst.h)
        self._hip = None
        self._hip_version = None
        self._hip_train = None
        # training (grad enabled, model.train()):
        #   "hip" (default; "hip-strict" is an alias) = conv trunk forward + backward in libmst.so (see _HipTrunk); a call the
        #                    hand-written trunk cannot take raises RuntimeError naming the reason -- nothing falls back silently
        #   "torch"        = explicit opt-in: everything on PyTorch-ROCm autograd (BASELINE configs[1] / A-B comparisons)
        #   "hip-or-torch" = explicit opt-in: as "hip", but a refused call warns once per reason and runs on PyTorch-ROCm autograd
        self.train_backend = "hip"
        # precision of the hand-written training trunk: "fp32" (exact), "f16" (float16 operands, fp32 accumulation:
        # the reference's --use_amp arithmetic, see include/mst.h mst_encoder_set_train_precision), "f16x3" (the same
        # kernels with 3-term split-precision operands: fp32-equivalent), or "auto" = f16 inside `torch.autocast(dtype=float16)`
        # (what src/train.py:251 turns on), fp32 otherwise
        self.train_precision = "auto"
        # data-parallel training: False = every rank normalises with the batch statistics of ITS clips (what
        # DistributedDataParallel without SyncBatchNorm does); True = the ranks of the default torch.distributed group add up
        # their statistics (exact integer sums over RCCL, 4 small all-reduces per step) and an N-rank step computes what the
        # single-process reference computes on the whole batch; a DistSync-like object = that, over its group
        self.sync_bn = False
        self._warned = set()
        self._stats_epoch = 0   # bumped whenever a kernel rewrites BatchNorm buffers through raw pointers (no tensor._version bump)

    def _params_version(self):
        return tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers()) + (self._stats_epoch,)

    def hip_encoder(self) -> HipEncoder:
        v = self._params_version() + (self.conv1_precision,)
        if self._hip is not None and v != self._hip_version and self._hip_version[-1] == self.conv1_precision and \
                self.film_encoder.film_head.weight.is_cuda and getattr(self._hip, "_dev", None) == self.film_encoder.film_head.weight.device:
            # weights changed (optimizer steps, load_state_dict): refresh the existing handle's tables on the device
            self._hip.update_from(self)
            self._hip_version = v
        elif self._hip is None or v != self._hip_version:   # first use / another precision mode / another device: build (host-side tables)
            self._hip, self._hip_version = HipEncoder(self, self.conv1_precision), v
            self._hip._dev = self.film_encoder.film_head.weight.device
        return self._hip

    def _needs_autograd(self, mixing_features):
        return torch.is_grad_enabled() and (self.training or mixing_features.requires_grad) and \
            any(p.requires_grad for p in self.parameters())

    _TRUNK_FAMILIES = (("conv1", "weight"), ("conv1", "bias"), ("bn1", "weight"), ("bn1", "bias"),
                       ("conv2", "weight"), ("conv2", "bias"), ("bn2", "weight"), ("bn2", "bias"))

    def _trunk_flat(self):
        """The conv / BatchNorm parameters of the sub-band CNNs as views of 8 STACKED tensors (one storage per family, stacked
        over the sub-bands): the training kernels read all sub-bands of a family as one tensor, and with the Parameters living
        inside it no torch.stack (11 copies per family, forward) and no per-band gradient copies (backward) run per step.
        Parameter objects, names, shapes and state_dict are unchanged; `.to()` / re-materialisation is detected by pointer and the
        storage is re-stacked.  Returns (stacked tensors, Parameters family-major)."""
        cn = self.audio_encoder.subnet_cnns
        params = [getattr(getattr(c, m), a) for m, a in self._TRUNK_FAMILIES for c in cn]
        flat = getattr(self, "_trunk_flat_bufs", None)
        ns = len(cn)
        ok = flat is not None and len(flat) == 8 and all(
            b.shape[0] == ns and params[f * ns].data_ptr() == b.data_ptr() and
            params[f * ns + ns - 1].data_ptr() == b[ns - 1].data_ptr() and params[f * ns].dtype == b.dtype
            for f, b in enumerate(flat))
        if not ok:   # ONE storage for the eight families (a pass's parameter snapshot is then a single copy), each family a view
            flat = []
            with torch.no_grad():
                fams = [params[f * ns:(f + 1) * ns] for f in range(8)]
                sizes = [ns * ps[0].numel() for ps in fams]
                base = torch.empty(sum(sizes), dtype=fams[0][0].dtype, device=fams[0][0].device)
                o = 0
                for ps, n in zip(fams, sizes):
                    buf = base[o:o + n].view((ns,) + tuple(ps[0].shape))
                    torch.stack([q.detach() for q in ps], out=buf)
                    for i, q in enumerate(ps):
                        q.data = buf[i]
                    flat.append(buf)
                    o += n
            self._trunk_flat_bufs = flat
        return flat, params

    def _bn_flat(self, name):
        """BatchNorm running statistics of layer `name` ("bn1" / "bn2") as views of 3 STACKED tensors (running_mean, running_var
        (n_sub, C) and num_batches_tracked (n_sub,)), so that the per-step update is five small launches per layer instead of one
        per sub-band and buffer.  Buffer objects, names, shapes and state_dict are unchanged (see _trunk_flat)."""
        bns = [getattr(c, name) for c in self.audio_encoder.subnet_cnns]
        cache = self.__dict__.setdefault("_bn_flat_bufs", {})
        flat = cache.get(name)
        ns = len(bns)
        keys = ("running_mean", "running_var", "num_batches_tracked")
        ok = flat is not None and all(
            b.shape[0] == ns and getattr(bns[0], k).data_ptr() == b.data_ptr() and getattr(bns[-1], k).data_ptr() == b[ns - 1].data_ptr()
            and getattr(bns[0], k).dtype == b.dtype for k, b in zip(keys, flat))
        if not ok:
            flat = []
            with torch.no_grad():
                for k in keys:
                    buf = torch.stack([getattr(bn, k).detach() for bn in bns])
                    for i, bn in enumerate(bns):
                        getattr(bn, k).data = buf[i]
                    flat.append(buf)
            cache[name] = flat
        return flat

    def _forward_train_hip(self, logmel, mixing_features):
        """Training forward with the trunk in libmst.so (see _HipTrunk); FiLM MLP and attention head stay torch modules."""
        ae, fe = self.audio_encoder, self.film_encoder
        enc = self._train_encoder()
        if isinstance(logmel, LogMel) and logmel.layout != enc.train_layout():
            logmel = logmel.to_reference()   # (a caller's own LogMel in a layout this precision mode does not read)
        cn = ae.subnet_cnns
        trunk_flat, trunk_params = self._trunk_flat()
        hip_small = self._small_nets_in_hip(mixing_features, (logmel.frames if isinstance(logmel, LogMel) else logmel.shape[-1]) // 20)
        if hip_small:   # FiLM MLP forward / backward in libmst.so (csrc/head.hip)
            mlp = fe.feature_mlp
            flat = _HipFilmMLP.apply(mixing_features, float(mlp[2].p) if self.training else 0.0, mlp[0].weight, mlp[0].bias,
                                     mlp[3].weight, mlp[3].bias, fe.film_head.weight, fe.film_head.bias)
        else:
            flat = fe.film_head(fe.feature_mlp(mixing_features))
        p = cn[0].dropout1.p if self.training else 0.0
        sync = None
        if self.sync_bn is True:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                sync = DistSync()
        elif self.sync_bn:
            sync = self.sync_bn
        # running statistics, as nn.BatchNorm2d updates them in training mode (unbiased variance): one launch per layer from the batch
        # statistics in the pass's workspace when the sub-bands share one momentum (they do in the reference), else per module
        mom = {name: {bn.momentum if bn.momentum is not None else 0.1 for bn in (getattr(c, name) for c in cn)} for name in ("bn1", "bn2")}
        fused_stats = all(len(v) == 1 for v in mom.values())
        out = _HipTrunk.apply(enc, logmel, flat, trunk_flat, float(p), sync, getattr(self, "_grad_reducer", None), not fused_stats,
                              *trunk_params)
        B, Fr = (logmel.B, logmel.frames) if isinstance(logmel, LogMel) else (logmel.shape[0], logmel.shape[-1])
        with torch.no_grad():
            if fused_stats:
                pool_in = out
                for layer, name in ((1, "bn1"), (2, "bn2")):
                    rm, rv, nb = self._bn_flat(name)
                    enc.update_running_stats(layer, B, Fr, rm, rv, nb, next(iter(mom[name])), cross_rank=sync is not None)
                self._stats_epoch += 1   # the kernel wrote the buffers behind autograd's back: the eval encoder's BatchNorm fold is stale
            else:
                pool_in, bn1, bn2 = out
                if sync is not None:   # the GLOBAL clip count, as the ranks summed it next to the statistics (ranks may hold different
                    B = enc.stats_view(1, B, Fr)[-2].to(torch.float64)   # numbers of clips); a device scalar: no host sync
                for stat, name, n in ((bn1, "bn1", B * (ae.split_size * Fr)), (bn2, "bn2", B * ((ae.split_size // enc.sub) * (Fr // 5)))):
                    corr = (n / torch.clamp(n - 1, min=1)).float() if torch.is_tensor(n) else n / max(n - 1, 1)
                    mean, var = stat[..., 0], (1.0 / stat[..., 1] ** 2 - cn[0].bn1.eps) * corr
                    for i, bn in enumerate(getattr(c, name) for c in cn):
                        m = bn.momentum if bn.momentum is not None else 0.1
                        bn.running_mean.mul_(1 - m).add_(mean[i], alpha=m)
                        bn.running_var.mul_(1 - m).add_(var[i], alpha=m)
                        bn.num_batches_tracked += 1
        if hip_small:   # Dropout + attention pooling + projection, forward / backward in libmst.so
            ap = ae.attention_pooling
            return _HipHead.apply(pool_in, float(cn[0].dropout2.p) if self.training else 0.0,
                                  float(ap.projection[2].p) if self.training else 0.0, ap.attention[0].weight, ap.attention[0].bias,
                                  ap.attention[2].weight, ap.attention[2].bias, ap.projection[0].weight, ap.projection[0].bias)
        x = F.dropout(pool_in, cn[0].dropout2.p, self.training)
        return ae.attention_pooling(x)

    def _train_encoder(self):
        """The training-side encoder handle with the precision mode this call resolves to (train_precision; "auto" = f16 inside
        torch.autocast(float16), fp32 otherwise)."""
        ae = self.audio_encoder
        if self._hip_train is None:
            self._hip_train = HipEncoder(self, "fp32")
        enc = self._hip_train
        if self.train_precision not in ("fp32", "f16", "f16x3", "auto"):
            raise ValueError("train_precision must be 'fp32', 'f16', 'f16x3' or 'auto'")
        want = self.train_precision
        if want == "auto":
            amp16 = torch.is_autocast_enabled() and (torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype")
                                                     else torch.get_autocast_gpu_dtype()) == torch.float16
            want = "f16" if amp16 else "fp32"
        if want != "fp32" and enc.sub != 2 and ae.split_size % 2:
            why = f"split_size={ae.split_size}: the f16 training kernels need an even split_size; the trunk stays fp32"
            if self.train_precision != "auto":
                raise RuntimeError(f"MixingStyleEncoder.train_precision='{self.train_precision}': " + why)
            if why not in self._warned:
                self._warned.add(why)
                warnings.warn("MixingStyleEncoder under autocast: " + why, RuntimeWarning, stacklevel=3)
            want = "fp32"
        enc.set_train_precision(want)
        return enc

    # the pooling head and the FiLM MLP of the training step: "hip" (default) = hand-written forward / backward (csrc/head.hip),
    # "torch" = the nn.Modules with autograd
    small_nets_backend = "hip"

    def _small_nets_in_hip(self, mixing_features, pooled_frames=1):
        ap, fe = self.audio_encoder.attention_pooling, self.film_encoder
        return (self.small_nets_backend == "hip" and not mixing_features.requires_grad and mixing_features.is_cuda
                and ap.attention[0].out_features <= 256 and ap.input_dim <= 3072 and fe.feature_dim <= 2048
                and pooled_frames <= 2048   # csrc/head.hip kHeadMaxFrames (LDS rows of the softmax / pooling kernels)
                and fe.feature_mlp[0].out_features <= 2048
                and all(q.dtype == torch.float32 for q in (ap.attention[0].weight, fe.film_head.weight)))

    _HIP_TRAIN_BACKENDS = ("hip", "hip-strict", "hip-or-torch")

    def _hip_trunk_refusal_for(self, frames, on_gpu, logmel_dtype=torch.float32):
        """Why the hand-written training trunk cannot take a call with this many frames (None = it can).  ONE predicate for
        `forward` (which must choose stage A's output layout before the log-mel exists) and `forward_from_logmel`."""
        if self.audio_encoder.split_size // 10 not in (1, 2):
            return f"split_size={self.audio_encoder.split_size}: the training kernels cover first-pool heights 1 and 2"
        if not on_gpu:
            return "log-mel is not on the GPU"
        if frames < 20:
            return f"{frames} frames < 20"
        if logmel_dtype != torch.float32 or self.film_encoder.film_head.weight.dtype != torch.float32:
            return f"non-fp32 tensors (log-mel {logmel_dtype}, parameters {self.film_encoder.film_head.weight.dtype})"
        return None

    def _hip_trunk_refusal(self, logmel):
        cm = isinstance(logmel, LogMel)
        return self._hip_trunk_refusal_for(logmel.frames if cm else logmel.shape[-1], (logmel.data if cm else logmel).is_cuda,
                                           torch.float32 if cm else logmel.dtype)

    def forward_from_logmel(self, logmel, mixing_features):
        if self.train_backend not in self._HIP_TRAIN_BACKENDS + ("torch",):
            raise ValueError("train_backend must be 'hip' (default; alias 'hip-strict'), 'torch' or 'hip-or-torch'")
        auto = self._needs_autograd(mixing_features)
        hip_train = self.encoder_backend == "hip" and self.training and auto and self.train_backend in self._HIP_TRAIN_BACKENDS
        if hip_train:   # decided BEFORE any layout conversion, so that a refusal names its reason
            why = self._hip_trunk_refusal(logmel)
            if why is None:
                return self._forward_train_hip(logmel, mixing_features)
            msg = f"MixingStyleEncoder: the hand-written HIP training trunk cannot take this call ({why})"
            if self.train_backend != "hip-or-torch":
                raise RuntimeError(msg + "; train_backend='torch' (or 'hip-or-torch') runs the conv stack on PyTorch-ROCm/MIOpen "
                                         "autograd instead (about 2x the step time, library numerics)")
            if self.sync_bn:   # the decision is per rank: a rank that fell back would skip the statistics all-reduces its peers
                raise RuntimeError(msg + " -- refused with sync_bn: the other ranks would wait in their collectives forever")
            if why not in self._warned:
                self._warned.add(why)
                warnings.warn(msg + "; running the conv stack on PyTorch-ROCm/MIOpen autograd (train_backend='hip-or-torch')",
                              RuntimeWarning, stacklevel=2)
        if isinstance(logmel, LogMel) and not (self.encoder_backend == "hip" and not auto and not self.training):
            logmel = logmel.to_reference()   # only the eval forward and the hand-written training trunk read channel-minor layouts
        if auto and self.training and self.sync_bn and "sync_bn" not in self._warned:
            self._warned.add("sync_bn")
            warnings.warn("MixingStyleEncoder.sync_bn is implemented by the hand-written HIP trunk only; this call runs the conv "
                          "stack on PyTorch-ROCm autograd with per-rank BatchNorm statistics (convert the modules with "
                          "torch.nn.SyncBatchNorm.convert_sync_batchnorm for the same semantics there)", RuntimeWarning, stacklevel=2)
        if self.encoder_backend == "hip" and not auto:
            if self.training:
                raise RuntimeError("HIP encoder forward implements eval-mode BatchNorm/Dropout; call model.eval() "
                                   "or enable grad for the training path")
            return self.hip_encoder().forward(logmel, mixing_features)
        return self.audio_encoder.forward_from_logmel(logmel, self.film_encoder(mixing_features))

    def forward(self, stems_dict, mixing_features):
        """reference src/model.py:508-542.  ONE stage-A launch over the waveform yields the log-mel and the mixing
        features; rows of `mixing_features` that are deferred placeholders (what the Dataset hands out from fork'd
        workers, mixing_utils.FEATURES_DEFERRED) are filled from it on the device -- no host sync -- and rows that
        hold real values are used as given."""
        bins = detailed_bins_for_feature_dim(self.film_encoder.feature_dim)
        pre = self.audio_encoder.mel_preprocessor
        has_feats = bins is not None and (bins == 0 or bins <= self.audio_encoder.n_mels)
        plan = pre.plan(bins if has_feats else 0)
        # the eval forward in libmst.so takes the log-mel in its internal channel-minor layout when stage A can write it
        # (whole-line stores there, 256-byte runs / ready-made float16 operands for conv1), and so does the hand-written
        # training trunk in its float16 modes; every other path -- fp32 training, the PyTorch backend -- gets the reference's
        # (B, 8, n_mels, frames) tensor
        layout, want_lo = _lib.LOGMEL_REF, True
        auto = self._needs_autograd(mixing_features)
        if self.encoder_backend == "hip" and not self.training and not auto:
            want = self.hip_encoder().preferred_layout()
            if plan.supports_layout(want):
                layout, want_lo = want, self.hip_encoder().mode != 3   # (plain float16: the high parts alone)
        elif self.encoder_backend == "hip" and self.training and auto and self.train_backend in self._HIP_TRAIN_BACKENDS and \
                self._hip_trunk_refusal_for(1 + next(iter(stems_dict.values())).shape[-1] // pre.hop_length,
                                            next(iter(stems_dict.values())).is_cuda) is None:
            # the float16 training trunk reads stage A's float16 planes (conv1 forward and its weight gradient); the SAME predicate
            # as forward_from_logmel's, so a high-parts-only LogMel is only ever produced for a call the trunk will take
            want = self._train_encoder().train_layout()
            if plan.supports_layout(want):
                layout, want_lo = want, self._train_encoder().train_mode != 1   # (f16 mode: the high parts alone)
        with torch.no_grad():
            logmel, feats = plan.forward_stems(stems_dict, True, has_feats, layout, want_absmax=layout == _lib.LOGMEL_CM16, want_lo=want_lo)
        mf = mixing_features.to(logmel.device)
        if feats is not None:
            mf = torch.where(is_deferred(mf), feats.to(mf.dtype), mf)
        elif bool(is_deferred(mf).any()):
            # no extractor layout of this size (an unusual feature_dim), so nothing can fill a placeholder: refuse, as a Python
            # error the caller can catch.  This branch alone reads one flag back; the extractor layouts above never sync.
            raise ValueError(f"mixing_features holds deferred placeholder rows but feature_dim={self.film_encoder.feature_dim} "
                             "is not a MixingFeatureExtractor layout")
        return self.forward_from_logmel(logmel, mf)
