"""MI355X mirror of reference src/mixing_utils.py: MixingFeatureExtractor and AudioAugmenter.

Same class names, constructor arguments, method names, return shapes and error behaviour as the
reference; the arithmetic runs in libmst.so (HIP, gfx950).  Beyond the reference API the
extractor is batched: `extract_all_features` accepts stems of shape (2, T) -> (feature_dim,)
as the reference does, or (B, 2, T) -> (B, feature_dim), and `features_and_logmel` returns the
model's log-mel input from the same pass over the waveform.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

STEMS = ("vocals", "bass", "drums", "other")

# Deferred mixing features.  The reference computes the features in the Dataset, inside fork'd DataLoader workers on the
# CPU (src/data.py:231-265); a HIP context cannot be used after fork, so a worker hands out a PLACEHOLDER row instead --
# a real (feature_dim,) fp32 tensor, every element FEATURES_DEFERRED -- and `MixingStyleEncoder.forward` fills such rows
# on the device from the same stage-A launch that produces the model's log-mel (one pass over the waveform).  Real
# features are clamped to [-100, 100] (src/mixing_utils.py:338-341), so the marker cannot collide with a value; it is
# finite (train.py:237's isnan check stays quiet) and exact in fp32 / fp16 / bf16.
FEATURES_DEFERRED = -32768.0


def deferred_features(feature_dim: int) -> torch.Tensor:
    return torch.full((feature_dim,), FEATURES_DEFERRED, dtype=torch.float32)


def is_deferred(features: torch.Tensor) -> torch.Tensor:
    """(..., F) -> (..., 1) bool: rows that are placeholders (decided on the tensor's own device, no host sync)."""
    return features[..., :1] == FEATURES_DEFERRED


def detailed_bins_for_feature_dim(feature_dim: int):
    """Inverse of MixingFeatureExtractor.get_feature_dim (src/mixing_utils.py:53-69): 4*(6+sd+3)+8 with sd = 5 (default)
    or n_spectral_bins + 2 (detailed mode).  -> n_spectral_bins (0 = default layout) or None if no layout has this size."""
    if feature_dim < 8 or (feature_dim - 8) % 4:
        return None
    sd = (feature_dim - 8) // 4 - 9
    if sd == 5:
        return 0
    return sd - 2 if sd >= 3 else None


def hann_window(n_fft: int) -> torch.Tensor:
    return torch.hann_window(n_fft, periodic=True, dtype=torch.float32)


def melscale_fbanks_htk(n_freqs: int, n_mels: int, sample_rate: int) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(f_min=0, f_max=sr//2, norm=None, mel_scale='htk') in fp32 on the host
    (the kernels consume this table; it is never re-derived on the device)."""
    import math
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    mel_max = 2595.0 * math.log10(1.0 + float(sample_rate // 2) / 700.0)
    mel_pts = torch.linspace(0.0, mel_max, n_mels + 2)
    hz = 700.0 * (10.0 ** (mel_pts / 2595.0) - 1.0)
    dhz = hz[1:] - hz[:-1]
    slope = hz.unsqueeze(0) - freqs.unsqueeze(1)
    lower = (-1.0 * slope[:, :-2]) / dhz[:-1]
    upper = slope[:, 2:] / dhz[1:]
    return torch.max(torch.zeros(1), torch.min(lower, upper))


class LogMel:
    """A log-mel batch in one of the ENCODER-INTERNAL channel-minor layouts that stage A writes directly (include/mst.h
    MST_LOGMEL_CM32 / CM16): `data` (B, frames, n_mels, 8) fp32, or float16 high parts with the low parts in `lo`;
    `absmax` (B,) int32 = max |log-mel| per clip as float bits (the range bound of the float16 convolutions).  `lo` may be
    None when only the plain-float16 kernels will read it (they use the high parts alone).  `HipEncoder.forward` and the float16
    training trunk consume it; `to_reference()` gives the reference's (B, 8, n_mels, frames) tensor."""

    def __init__(self, layout, data, lo=None, absmax=None):
        self.layout, self.data, self.lo, self.absmax = layout, data, lo, absmax
        self.B, self.frames, self.n_mels = data.shape[0], data.shape[1], data.shape[2]
        self.device = data.device

    def to_reference(self):
        if self.layout == _lib.LOGMEL_CM16 and self.lo is None:
            raise _lib.MstError("this LogMel holds the float16 HIGH parts only (what the plain-float16 kernels read): the fp32 "
                                "log-mel cannot be recovered from it -- ask stage A for the reference layout instead")
        x = self.data.float() if self.lo is None else self.data.float() + self.lo.float()
        return x.permute(0, 3, 2, 1).contiguous()


class MelFeatPlan:
    """Owns an `mst_plan` (device tables for one STFT/mel configuration) and a cached workspace."""

    def __init__(self, sample_rate, n_fft, hop_length, n_mels, detailed_bins=0, window=None, fb=None):
        self.sample_rate, self.n_fft, self.hop_length, self.n_mels = sample_rate, n_fft, hop_length, n_mels
        self.window = (hann_window(n_fft) if window is None else window).detach().float().cpu().contiguous()
        self.fb = (melscale_fbanks_htk(n_fft // 2 + 1, n_mels, sample_rate) if fb is None else fb)
        self.fb = self.fb.detach().float().cpu().contiguous()
        h = C.c_void_p()
        L = _lib.lib()
        _lib.check(L.mst_plan_create(C.byref(h), sample_rate, n_fft, hop_length, n_mels,
                                     C.c_void_p(self.window.data_ptr()), C.c_void_p(self.fb.data_ptr()),
                                     int(detailed_bins)), "mst_plan_create")
        self._h = h
        self.feature_dim = L.mst_plan_feature_dim(h)
        self._ws = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().mst_plan_destroy(h)
            except Exception:
                pass

    def frames(self, T: int) -> int:
        return 1 + T // self.hop_length

    def _workspace(self, B, T, device):
        need = _lib.lib().mst_melfeat_workspace_bytes(self._h, B, T)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws, need

    def supports_layout(self, layout) -> bool:
        """Whether stage A writes `layout` (_lib.LOGMEL_*) directly for this configuration."""
        return bool(_lib.lib().mst_plan_layout_supported(self._h, int(layout)))

    def _run(self, ptrs4, stride, pcm16, B, T, dev, want_logmel, want_feats, layout, want_absmax, want_lo=True):
        F = self.frames(T)
        logmel = lo = absmax = None
        if want_logmel:
            if layout == _lib.LOGMEL_REF:
                logmel = torch.empty(B, 8, self.n_mels, F, dtype=torch.float32, device=dev)
            elif layout == _lib.LOGMEL_CM32:
                logmel = torch.empty(B, F, self.n_mels, 8, dtype=torch.float32, device=dev)
            else:
                logmel = torch.empty(B, F, self.n_mels, 8, dtype=torch.float16, device=dev)
                lo = torch.empty_like(logmel) if want_lo else None   # (plain-float16 consumers read the high parts only)
            if want_absmax and layout != _lib.LOGMEL_REF:
                absmax = torch.empty(B, dtype=torch.int32, device=dev)
        feats = torch.empty(B, self.feature_dim, dtype=torch.float32, device=dev) if want_feats else None
        ws, need = self._workspace(B, T, dev)
        io = _lib.MelfeatIO((C.c_void_p * 4)(*ptrs4), stride, int(pcm16), int(layout), _lib.dptr(logmel), _lib.dptr(lo),
                            _lib.dptr(absmax), _lib.dptr(feats))
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mst_melfeat_forward_io(self._h, C.byref(io), B, T, _lib.dptr(ws), need,
                                                          _lib.stream_ptr(dev)), "mst_melfeat_forward_io")
        if logmel is not None and layout != _lib.LOGMEL_REF:
            logmel = LogMel(layout, logmel, lo, absmax)
        return logmel, feats

    def forward_stems(self, stems_dict, want_logmel=True, want_feats=True, layout=_lib.LOGMEL_REF, want_absmax=False, want_lo=True):
        """{stem: (B,2,T) | (2,T)} fp32 CUDA -> (logmel, feats) without concatenating the stems: the kernel reads the
        four tensors in place (views of one packed (B,8,T) tensor work too, any common clip stride).
        layout: _lib.LOGMEL_REF -> the reference's (B, 8, n_mels, frames) tensor; LOGMEL_CM32 / CM16 -> a `LogMel` in the
        encoder-internal channel-minor layout (check `supports_layout` first)."""
        parts = [stems_dict[s] for s in STEMS]
        if parts[0].dim() == 2:
            parts = [q.unsqueeze(0) for q in parts]
        if not parts[0].is_cuda:
            raise _lib.MstError("libmst kernels need CUDA (HIP) tensors; got a CPU tensor and there is no CPU fallback")
        B, ch, T = parts[0].shape
        dt = parts[0].dtype
        ok = all(q.dtype == dt and tuple(q.shape) == (B, 2, T) and q.stride(2) == 1 and q.stride(1) == T
                 and q.stride(0) == parts[0].stride(0) for q in parts) and (B == 1 or parts[0].stride(0) >= 2 * T) \
            and dt in (torch.float32, torch.int16)
        if not ok:
            if all(q.dtype == torch.int16 for q in parts):
                return self.forward(torch.cat(parts, dim=1), want_logmel, want_feats, layout, want_absmax, want_lo)
            return self.forward(torch.cat([q.float() for q in parts], dim=1), want_logmel, want_feats, layout, want_absmax, want_lo)
        stride = parts[0].stride(0) if B > 1 else 2 * T
        return self._run([q.data_ptr() for q in parts], stride, dt == torch.int16, B, T, parts[0].device, want_logmel,
                         want_feats, layout, want_absmax, want_lo)

    def forward(self, stems8: torch.Tensor, want_logmel=True, want_feats=True, layout=_lib.LOGMEL_REF, want_absmax=False, want_lo=True):
        """stems8 (B, 8, T) CUDA, fp32 or int16 PCM (value = s / 32768) -> (logmel (B,8,M,F) | LogMel | None, feats (B,Fd) | None)."""
        if not stems8.is_cuda:
            raise _lib.MstError("libmst kernels need CUDA (HIP) tensors; got a CPU tensor and there is no CPU fallback")
        pcm16 = stems8.dtype == torch.int16
        x = stems8.contiguous() if pcm16 else stems8.contiguous().float()
        B, ch, T = x.shape
        assert ch == 8, "expected 8 channels (4 stems x stereo)"
        if layout == _lib.LOGMEL_REF and not want_absmax:   # the packed-tensor entry points of the C ABI
            F = self.frames(T)
            logmel = torch.empty(B, 8, self.n_mels, F, dtype=torch.float32, device=x.device) if want_logmel else None
            feats = torch.empty(B, self.feature_dim, dtype=torch.float32, device=x.device) if want_feats else None
            ws, need = self._workspace(B, T, x.device)
            L = _lib.lib()
            fn = L.mst_melfeat_forward_pcm16 if pcm16 else L.mst_melfeat_forward
            with torch.cuda.device(x.device):
                _lib.check(fn(self._h, _lib.dptr(x), B, T, _lib.dptr(logmel), _lib.dptr(feats), _lib.dptr(ws), need,
                              _lib.stream_ptr(x.device)), "mst_melfeat_forward")
            return logmel, feats
        es = x.element_size()
        return self._run([x.data_ptr() + 2 * s * T * es for s in range(4)], 8 * T, pcm16, B, T, x.device, want_logmel,
                         want_feats, layout, want_absmax, want_lo)


def stems_to_tensor(stems_dict) -> torch.Tensor:
    """{stem: (2,T) | (B,2,T)} -> (B, 8, T) in channel order vL,vR,bL,bR,dL,dR,oL,oR."""
    parts = [stems_dict[s] for s in STEMS]
    if parts[0].dim() == 2:
        parts = [p.unsqueeze(0) for p in parts]
    return torch.cat(parts, dim=1)


class _MelTransformView:
    """Attribute surface of torchaudio.transforms.MelSpectrogram that reference callers touch
    (`.spectrogram.window.device`, `.to()`; src/mixing_utils.py:155-156,274-275)."""

    class _Ns:
        pass

    def __init__(self, window, fb):
        self.spectrogram = self._Ns()
        self.spectrogram.window = window
        self.mel_scale = self._Ns()
        self.mel_scale.fb = fb

    def to(self, device):
        self.spectrogram.window = self.spectrogram.window.to(device)
        self.mel_scale.fb = self.mel_scale.fb.to(device)
        return self


class MixingFeatureExtractor:
    """Extract interpretable mixing features from audio stems (reference src/mixing_utils.py:16-357)."""

    def __init__(self, sample_rate=44100, n_fft=1024, hop_length=256, n_mels=128, use_detailed_spectral=False,
                 n_spectral_bins=32):
        self.sr = sample_rate
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.n_mels = n_mels
        self.use_detailed_spectral = use_detailed_spectral
        self.n_spectral_bins = n_spectral_bins
        self._plan = None
        self.mel_transform = _MelTransformView(hann_window(n_fft), melscale_fbanks_htk(n_fft // 2 + 1, n_mels,
                                                                                      sample_rate))

    def get_feature_dim(self):
        spectral = 5 if not self.use_detailed_spectral else (self.n_spectral_bins + 2)
        return 4 * (6 + spectral + 3) + 4 + 4

    def plan(self) -> MelFeatPlan:
        if self._plan is None:
            self._plan = MelFeatPlan(self.sr, self.n_fft, self.hop_length, self.n_mels,
                                     self.n_spectral_bins if self.use_detailed_spectral else 0)
        return self._plan

    def features_and_logmel(self, stems_dict, layout=_lib.LOGMEL_REF, want_absmax=False):
        """One pass over the waveform: returns (features (B,Fd), logmel (B,8,M,F)); with a channel-minor `layout`
        (_lib.LOGMEL_CM32 / CM16) the log-mel is a `LogMel` in the encoder-internal layout instead."""
        lm, f = self.plan().forward_stems(stems_dict, True, True, layout, want_absmax)
        return f, lm

    def resolve_features(self, stems_dict, features):
        """Fill the placeholder rows of `features` (see FEATURES_DEFERRED) from the stems, on the stems' device."""
        f = self.extract_all_features(stems_dict)
        features = features.to(f.device)
        return torch.where(is_deferred(features), f, features.to(f.dtype))

    def extract_all_features(self, stems_dict):
        """stems_dict {stem: (2,T)} -> (feature_dim,)  [reference]; {stem: (B,2,T)} -> (B, feature_dim)."""
        batched = next(iter(stems_dict.values())).dim() == 3
        _, f = self.plan().forward_stems(stems_dict, False, True)
        return f if batched else f[0]

    # ---- per-group views of the fused feature vector (reference public sub-methods)
    def _single(self, audio):
        z = torch.zeros_like(audio)
        f = self.extract_all_features({"vocals": audio, "bass": z, "drums": z, "other": z})
        sd = 5 if not self.use_detailed_spectral else self.n_spectral_bins + 2
        return f, 3 * (10 + sd) + 4, sd

    def extract_dynamics(self, audio):
        f, o, _ = self._single(audio)
        return f[o:o + 6]

    def extract_spectral(self, audio):
        f, o, sd = self._single(audio)
        return f[o + 7:o + 7 + sd]

    def extract_stereo(self, audio):
        f, o, sd = self._single(audio)
        return f[o + 7 + sd:o + 10 + sd]

    def compute_loudness(self, audio):
        f, o, _ = self._single(audio)
        return f[o + 4]

    def extract_masking(self, stems_dict):
        f = self.extract_all_features(stems_dict)
        sd = 5 if not self.use_detailed_spectral else self.n_spectral_bins + 2
        return f[2 * (10 + sd):2 * (10 + sd) + 4]


# =============================================================================
# Audio augmentations (degradations)
# =============================================================================

class AudioAugmenter:
    """Apply mixing degradation augmentations to separated stems (reference src/mixing_utils.py:364-479).

    Every random decision is drawn on the host from the global torch CPU generator in exactly the reference's
    order -- per stem [coin, (gain)] [coin, (hi/lo coin)] [coin] [coin, (cutoff)], then [coin, (randn(L))] -- so a
    seeded run makes bit-identical decisions; the audio is then processed by libmst.so (`mst_aug_apply`).
    `augment_stems` accepts the reference's {stem: (2, T)} dict or a batched {stem: (B, 2, T)} dict (clips are drawn
    one after the other, as consecutive reference calls would).  `last_trace` keeps the decisions of the last call.
    """

    def __init__(self, sample_rate=44100, gain_range=9.0, prob=0.5):
        self.sr = sample_rate
        self.gain_range = gain_range
        self.prob = prob
        self._sos_cache = {}
        self._ws = None
        self.last_trace = None

    # -- host side: decisions ------------------------------------------------------------------
    def _butter(self, order, fc, btype):
        from scipy.signal import butter
        key = (order, float(fc), btype)
        if key not in self._sos_cache:
            if len(self._sos_cache) > 64:
                self._sos_cache.clear()
            self._sos_cache[key] = np.ascontiguousarray(butter(order, fc, btype=btype, fs=self.sr, output="sos"),
                                                        dtype=np.float64)
        return self._sos_cache[key]

    def _draw_tilt(self, st, tr):
        hi = bool(torch.rand(1) < 0.5)
        sos = self._butter(2, 2000, "high") if hi else self._butter(2, 500, "low")
        st.tilt = 1
        st.tilt_sos[:] = sos[0].tolist()
        tr["tilt"] = "high" if hi else "low"

    def _butter4_low(self, fc):
        """Closed-form butter(4, fc, 'low', fs=self.sr, output='sos') (bilinear transform of the two analog
        second-order sections, damping sin(pi/8) and cos(pi/8)); scipy.signal.butter costs ~0.6 ms per call on the
        host and the cutoff is a fresh random number per stem.  Agrees with scipy to ~1e-15 (tests/test_abi.py);
        scipy puts the whole gain into the first section, here every section is normalised -- same cascade."""
        t = np.tan(np.pi * fc / self.sr)
        sos = np.empty((2, 6), dtype=np.float64)
        for i, zeta in enumerate((np.cos(np.pi / 8.0), np.sin(np.pi / 8.0))):
            a0 = 1.0 + 2.0 * zeta * t + t * t
            g = t * t / a0
            sos[i] = (g, 2.0 * g, g, 1.0, 2.0 * (t * t - 1.0) / a0, (1.0 - 2.0 * zeta * t + t * t) / a0)
        return sos

    def _draw_bw(self, st, tr):
        cutoff = torch.rand(1) * 8000 + 4000
        sos = self._butter4_low(cutoff.item())
        st.bw_sections = sos.shape[0]
        st.bw_sos[:] = sos.reshape(-1).tolist()
        tr["cutoff"] = cutoff.item()

    def _make_ir(self, decay=0.5, trace=None):
        n = int(self.sr * decay)
        t = torch.linspace(0, decay, n)
        r = torch.randn(n)   # the reference's draw (src/mixing_utils.py:463), global torch CPU generator
        if trace is not None:
            trace["reverb_randn"] = r
        return torch.exp(-t / (decay / 4)) * r * 0.1

    def _draw_clip(self, clip):
        trace = {}
        for i, name in enumerate(STEMS):
            st, tr = clip.stem[i], {}
            st.gain = 1.0
            if torch.rand(1) < self.prob:
                gain_db = torch.rand(1) * 2 * self.gain_range - self.gain_range
                st.gain = (10 ** (gain_db / 20)).item()
                tr["gain_db"] = gain_db.item()
            if torch.rand(1) < self.prob:
                self._draw_tilt(st, tr)
            if torch.rand(1) < self.prob:
                st.compress = 1
                tr["comp"] = True
            if torch.rand(1) < self.prob:
                self._draw_bw(st, tr)
            trace[name] = tr
        ir = None
        if torch.rand(1) < self.prob:
            clip.reverb = 1
            ir = self._make_ir(trace=trace)
            trace["reverb_ir"] = ir
        return ir, trace

    # -- device side ---------------------------------------------------------------------------
    def _apply(self, x8, clips, irs, src=None):
        """x8 (B, 8, T) fp32 CUDA, modified in place; every clip's (8, T) block contiguous, clips any stride >= 8 T apart.
        src: read the audio from this (B, 8, T) tensor instead (never written; x8 is then output only, `mst_aug_apply_from`)."""
        if not x8.is_cuda:
            raise _lib.MstError("libmst kernels need CUDA (HIP) tensors; got a CPU tensor and there is no CPU fallback")
        B, _, T = x8.shape
        assert x8.dtype == torch.float32 and x8.stride(2) == 1 and x8.stride(1) == T and (B == 1 or x8.stride(0) >= 8 * T)
        L = _lib.lib()
        ir_len = max([0] + [ir.numel() for ir in irs if ir is not None])
        ir_dev = None
        if ir_len:   # impulse responses: pinned staging buffers (two, alternating: the previous call's copy may still be in
            #          flight) and an asynchronous copy on the call's stream -- a pageable `.to(device)` is a host synchronisation
            k = self._ir_turn = 1 - getattr(self, "_ir_turn", 0)
            bufs = self.__dict__.setdefault("_ir_bufs", [None, None])
            if bufs[k] is None or tuple(bufs[k][0].shape) != (B, ir_len) or bufs[k][1].device != x8.device:
                bufs[k] = (torch.zeros(B, ir_len).pin_memory(), torch.empty(B, ir_len, device=x8.device), torch.cuda.Event())
            host, ir_dev, done = bufs[k]
            done.synchronize()   # the copy that last read this pinned buffer (two calls ago)
            for b, ir in enumerate(irs):
                if ir is not None:
                    host[b] = ir
            with torch.cuda.device(x8.device):
                ir_dev.copy_(host, non_blocking=True)
                done.record()
        need = L.mst_aug_workspace_bytes(B, T, ir_len)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x8.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x8.device)
        with torch.cuda.device(x8.device):
            if src is None:
                _lib.check(L.mst_aug_apply_strided(clips, B, T, _lib.dptr(x8), x8.stride(0) if B > 1 else 8 * T, _lib.dptr(ir_dev),
                                                   ir_len, _lib.dptr(self._ws), need, _lib.stream_ptr(x8.device)),
                           "mst_aug_apply_strided")
            else:
                assert src.is_cuda and src.device == x8.device and src.dtype == torch.float32 and tuple(src.shape) == (B, 8, T)
                assert src.stride(2) == 1 and src.stride(1) == T and (B == 1 or src.stride(0) >= 8 * T)
                _lib.check(L.mst_aug_apply_from(clips, B, T, _lib.dptr(src), src.stride(0) if B > 1 else 8 * T, _lib.dptr(x8),
                                                x8.stride(0) if B > 1 else 8 * T, _lib.dptr(ir_dev), ir_len, _lib.dptr(self._ws),
                                                need, _lib.stream_ptr(x8.device)), "mst_aug_apply_from")
        return x8

    def draw_decisions(self, n_clips):
        """Draw the decisions of the next `n_clips` clips now (host RNG only; e.g. while the GPU is busy)."""
        clips = (_lib.AugClip * n_clips)()
        out = [self._draw_clip(clips[b]) for b in range(n_clips)]
        return clips, [o[0] for o in out], [o[1] for o in out]

    def augment_stems(self, stems_dict, decisions=None):
        batched = next(iter(stems_dict.values())).dim() == 3
        x8 = stems_to_tensor(stems_dict).float().contiguous()     # the reference's `.clone()`
        if x8.data_ptr() == next(iter(stems_dict.values())).data_ptr():
            x8 = x8.clone()
        B = x8.shape[0]
        clips, irs, traces = decisions if decisions is not None else self.draw_decisions(B)
        assert len(irs) == B
        self.last_trace = traces if batched else traces[0]
        self._apply(x8, clips, irs)
        out = {s: x8[:, 2 * i:2 * i + 2] for i, s in enumerate(STEMS)}
        return out if batched else {s: v[0] for s, v in out.items()}

    def augment_packed_(self, x8, decisions=None, src=None):
        """IN PLACE on a packed (B, 8, T) CUDA fp32 tensor (channels vL, vR, bL, bR, dL, dR, oL, oR) whose clips may be
        strided -- e.g. `batch[2::3]`, the negatives of a triplet batch, augmented where they stand (the caller has already
        made the copy the reference's `.clone()` stands for).  Same decisions / RNG order as `augment_stems`.
        src: a (B, 8, T) tensor in other memory to read the audio from -- x8 = augment(src), src untouched: the reference's
        `.clone()` (src/mixing_utils.py:386) rides in the chain's first pass instead of being a copy before it."""
        clips, irs, traces = decisions if decisions is not None else self.draw_decisions(x8.shape[0])
        assert len(irs) == x8.shape[0]
        self.last_trace = traces
        return self._apply(x8, clips, irs, src)

    def _single(self, audio, fill):
        """Run one effect on a (2, T) tensor: it rides as the first stem of an otherwise silent clip."""
        x8 = torch.zeros(1, 8, audio.shape[-1], dtype=torch.float32, device=audio.device)
        x8[0, 0:2] = audio
        clips = (_lib.AugClip * 1)()
        for i in range(4):
            clips[0].stem[i].gain = 1.0
        ir = fill(clips[0])
        self._apply(x8, clips, [ir])
        return x8[0, 0:2]

    def apply_spectral_tilt(self, audio):
        return self._single(audio, lambda c: self._draw_tilt(c.stem[0], {}))

    def apply_compression(self, audio, threshold=-20, ratio=4):
        """reference src/mixing_utils.py:435-447, any threshold (dB) and ratio; the default setting takes the closed form with two
        square roots (include/mst.h mst_aug_stem.compress = 1), any other one the general power law (compress = 2)."""
        if not ratio > 0:
            raise ValueError(f"apply_compression: ratio must be > 0, got {ratio}")

        def fill(c):
            st = c.stem[0]
            if threshold == -20 and ratio == 4:
                st.compress = 1
            else:
                st.compress, st.comp_threshold_db, st.comp_ratio = 2, float(threshold), float(ratio)
        return self._single(audio, fill)

    def apply_bandwidth_limit(self, audio):
        return self._single(audio, lambda c: self._draw_bw(c.stem[0], {}))

    def apply_reverb(self, audio, decay=0.5):
        def fill(c):
            c.reverb = 2   # plain reverb of the first stem: 0.7 * x + 0.3 * xcorr(x, ir)
            return self._make_ir(decay)
        return self._single(audio, fill)
