"""MI355X mirror of reference src/data.py (contrastive part): SCNetSeparator, FMABaselineDataset,
baseline_collate_fn -- same signatures, return tuples and exceptions.

Differences forced by the device path:
  * `compute_features`: the reference computes the 64-d features inside the (fork'd) DataLoader worker on the CPU
    (src/data.py:231-265).  A HIP context cannot be used after fork, so with the default `compute_features="deferred"`
    every features slot holds a PLACEHOLDER row -- a real `(feature_dim,)` fp32 tensor filled with
    `mixing_utils.FEATURES_DEFERRED` -- which survives `torch.stack`, `torch.isnan(...)`, `.to(device)`
    (src/train.py:237,243) and `features_list[0].shape[0]` (src/train.py:523) unchanged, and
    `MixingStyleEncoder.forward(stems_dict, mixing_features)` fills such rows on the device from the same stage-A
    launch that produces its log-mel (one pass over the waveform, no host sync).  `compute_features=True` computes
    the real features in `__getitem__` on the GPU (main process / spawn workers only).
  * stem decoding: the reference reads `{stem}.mp3` with torchaudio.load (not installed here).  Decoding stays a
    host concern: `stem_loader` is a pluggable callable `path -> (tensor (C, L), sample_rate)`;
    the default tries torchaudio, then a PCM `.wav` reader for pre-decoded stems.
SCNet source and weights are not part of the reference tree (un-vendored submodule): `SCNetSeparator` is
the interface only and needs a user-registered backend.
"""
import glob
import os
import wave

import numpy as np
import torch
from torch.utils.data import Dataset

from .mixing_utils import STEMS, MixingFeatureExtractor, deferred_features

_SEPARATOR_BACKEND = None


def register_separator_backend(factory):
    """factory(model_path, config_path, device) -> object with .separate(np.ndarray (2,T)) -> {stem: np.ndarray (2,T)}"""
    global _SEPARATOR_BACKEND
    _SEPARATOR_BACKEND = factory


class SCNetSeparator:
    """Wrapper for SCNet source separation (reference src/data.py:28-108): audio (2,T)|(T,) -> 4 stems (2,T) float CPU."""

    def __init__(self, model_path, config_path, device="cuda"):
        if _SEPARATOR_BACKEND is None:
            raise RuntimeError(
                "SCNetSeparator: the SCNet implementation (ZFTurbo/Music-Source-Separation-Training) and its checkpoint "
                "are not part of the reference tree (empty submodule); register a backend with "
                "mst_amd.data.register_separator_backend(factory) or use pre-separated stems (FMABaselineDataset).")
        self.device = device
        self._impl = _SEPARATOR_BACKEND(model_path, config_path, device)
        self.sample_rate = getattr(self._impl, "sample_rate", 44100)

    @torch.no_grad()
    def separate(self, audio):
        if isinstance(audio, torch.Tensor):
            audio = audio.cpu().numpy()
        if audio.ndim == 1:
            audio = np.stack([audio, audio], axis=0)
        out = self._impl.separate(audio)
        return {s: torch.from_numpy(np.asarray(out[s])).float() for s in STEMS}


def _load_wav(path):
    with wave.open(path, "rb") as w:
        ch, sw, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"unsupported PCM width {sw} in {path}")
    return torch.from_numpy(a.reshape(-1, ch).T.copy()), sr


def _is_riff(path):
    try:
        with open(path, "rb") as f:
            h = f.read(12)
        return h[:4] == b"RIFF" and h[8:12] == b"WAVE"
    except OSError:
        return False


def default_stem_loader(path):
    """torchaudio.load when torchaudio is installed (what the reference calls, src/data.py:171); otherwise PCM RIFF
    files are read directly -- recognised by CONTENT, so pre-decoded stems may keep the `{stem}.mp3` names the reference
    hard-codes (src/data.py:188).  Compressed audio without torchaudio is an error, never a silent fallback."""
    try:
        import torchaudio  # noqa: F401
        return torchaudio.load(path)
    except ImportError:
        if _is_riff(path):
            return _load_wav(path)
        raise RuntimeError(f"cannot decode {path}: torchaudio is not installed; pass stem_loader= or use PCM .wav stems")


def resample_sinc_hann(audio, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """`torchaudio.transforms.Resample(orig_freq, new_freq)(audio)` -- what the reference calls when a stem's rate differs from the
    dataset's (src/data.py:174-176, :425-427) -- restated from torchaudio's published algorithm for the case that torchaudio is not
    installed (torchaudio/functional/functional.py `_get_sinc_resample_kernel` / `_apply_sinc_resample_kernel`, defaults:
    `sinc_interp_hann`, lowpass_filter_width 6, rolloff 0.99): a bank of new_freq/gcd Hann-windowed sinc filters at
    0.99 x the lower Nyquist, evaluated in float64 and rounded to float32, applied as one strided conv1d; output length
    ceil(new * n / orig).  Host-side (a Dataset worker's job); pinned to the algorithm, not to a torchaudio build (absent here)."""
    import math

    import torch.nn.functional as F
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return audio
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
    kernels = kernels.to(torch.float32)
    shape = audio.shape
    wav = audio.reshape(-1, shape[-1]).float()
    n = wav.shape[-1]
    wav = F.pad(wav, (width, width + orig))
    out = F.conv1d(wav[:, None], kernels, stride=orig).transpose(1, 2).reshape(wav.shape[0], -1)
    out = out[..., :int(math.ceil(new * n / orig))]
    return out.reshape(shape[:-1] + out.shape[-1:])


def _clip_features(ds, clip_stems):
    """Features slot of one clip: the real vector (`compute_features=True`, needs a usable GPU context in this
    process) or the deferred placeholder row the model fills in on the device (default; "auto" is an alias)."""
    if ds.compute_features is True:
        dev = ds.device or "cuda"
        return ds.feature_extractor.extract_all_features({k: v.to(dev) for k, v in clip_stems.items()}).cpu()
    if ds.compute_features not in ("deferred", "auto", False, None):
        raise ValueError(f"compute_features={ds.compute_features!r}: expected True or 'deferred'")
    return deferred_features(ds.feature_extractor.get_feature_dim())


class FMABaselineDataset(Dataset):
    """Pre-separated stems, positive pairs = different temporal segments of one song (reference src/data.py:111-288)."""

    def __init__(self, separated_path="/nas/FMA/fma_separated/", clip_duration=10.0, sample_rate=44100, n_fft=1024,
                 hop_length=256, n_mels=128, num_segments=2, min_audio_duration=25.0, compute_features="deferred",
                 stem_loader=None, stem_ext=".mp3", device=None):
        self.separated_path = separated_path
        self.clip_duration = clip_duration
        self.sr = sample_rate
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.n_mels = n_mels
        self.clip_samples = int(clip_duration * sample_rate)
        self.num_segments = num_segments
        self.min_duration = min_audio_duration
        self.compute_features = compute_features
        self.stem_loader = stem_loader or default_stem_loader
        self.stem_ext = stem_ext
        self.device = device
        if not os.path.exists(separated_path):
            raise ValueError(f"Separated stems directory not found: {separated_path}")
        self.track_dirs = [d for d in glob.glob(os.path.join(separated_path, "*")) if os.path.isdir(d)]
        self.feature_extractor = MixingFeatureExtractor(sample_rate, n_fft, hop_length, n_mels)

    def __len__(self):
        return len(self.track_dirs)

    def _load_single_stem(self, stem_path):
        audio, sr = self.stem_loader(stem_path)
        audio = audio.float()
        if sr != self.sr:
            try:
                import torchaudio
                audio = torchaudio.transforms.Resample(sr, self.sr)(audio)
            except ImportError:
                audio = resample_sinc_hann(audio, sr, self.sr)   # torchaudio's published algorithm, restated
        if audio.shape[0] == 1:
            audio = audio.repeat(2, 1)
        elif audio.shape[0] > 2:
            audio = audio[:2, :]
        return audio

    def _load_stems(self, track_dir):
        stems = {}
        for name in STEMS:
            path = os.path.join(track_dir, f"{name}{self.stem_ext}")
            if not os.path.exists(path):
                raise FileNotFoundError(
                    f"Stem file not found: {path}\nTrack directory: {track_dir}\n"
                    f"This indicates the pre-separated stems are missing or in wrong format.")
            stems[name] = self._load_single_stem(path)
        return stems

    def _features(self, clip_stems):
        return _clip_features(self, clip_stems)

    def _crop_starts(self, audio_length):
        """numpy global-RNG draws in the reference's order (src/data.py:222-266)."""
        C = self.clip_samples
        if self.num_segments == 1:
            m = audio_length - C
            return [0 if m <= 0 else int(np.random.randint(0, m + 1))]
        if self.num_segments == 2:
            if audio_length < 2 * C:
                return [0, 0]
            s1 = int(np.random.randint(0, audio_length - 2 * C + 1))
            s2 = int(np.random.randint(s1 + C, audio_length - C + 1))
            return [s1, s2]
        raise ValueError(f"num_segments={self.num_segments} is not supported. "
                         f"Only num_segments=1 or num_segments=2 are implemented.")

    def __getitem__(self, idx):
        track_dir = self.track_dirs[idx]
        if self.num_segments not in (1, 2):
            self._crop_starts(0)  # raises the reference's ValueError
        stems_full = self._load_stems(track_dir)
        audio_length = stems_full["vocals"].shape[1]
        stems_list, features_list = [], []
        for start in self._crop_starts(audio_length):
            clip = self._extract_clip(stems_full, start, audio_length)
            stems_list.append(clip)
            features_list.append(self._features(clip))
        return stems_list, features_list, idx, track_dir

    def _extract_clip(self, stems_full, start_idx, audio_length):
        out = {}
        for name, audio in stems_full.items():
            seg = audio[:, start_idx:start_idx + self.clip_samples]
            if seg.shape[1] < self.clip_samples:
                seg = torch.nn.functional.pad(seg, (0, self.clip_samples - seg.shape[1]))
            out[name] = seg
        return out


def baseline_collate_fn(batch):
    """List of (stems_list, features_list, song_idx, track_dir) -> (stems_dict {stem: (N,2,T)}, features (N,F) (rows may
    be deferred placeholders, see mixing_utils.FEATURES_DEFERRED), song_labels (N,) int64, track_dirs [N]).
    reference src/data.py:291-328"""
    stems, feats, labels, dirs = [], [], [], []
    for stems_list, features_list, song_idx, track_dir in batch:
        for s, f in zip(stems_list, features_list):
            stems.append(s)
            feats.append(f)
            labels.append(song_idx)
            dirs.append(track_dir)
    stems_dict = {name: torch.stack([s[name] for s in stems], dim=0) for name in STEMS}
    return stems_dict, torch.stack(feats, dim=0), torch.tensor(labels, dtype=torch.long), dirs


def shard_batch(stems_dict, features, song_labels, rank, world_size):
    """Clip-shard a collated batch across ranks: rank r gets clips [r*N/W, (r+1)*N/W)."""
    n = song_labels.shape[0]
    lo, hi = rank * n // world_size, (rank + 1) * n // world_size
    return ({k: v[lo:hi] for k, v in stems_dict.items()}, None if features is None else features[lo:hi],
            song_labels[lo:hi])


class StyleTransferDataset(Dataset):
    """(input_stems, target_stems, target_features) with input and target from DIFFERENT songs -- reference
    src/data.py:331-538, on the same feature kernels.  numpy global-RNG draws in the reference's order:
    input crop `randint(0, L - C)` (only when L > C; upper bound exclusive, :481-482), target index `randint(0, len)`
    redrawn while equal to idx (:516-519), target crop.  Short tracks are zero-padded (:471-477).
    `compute_features` as in FMABaselineDataset: by default the target-features slot is a deferred placeholder row
    that `MixingStyleEncoder.forward(target_stems, target_features)` (src/train_style_transfer.py:206-211) fills in on
    the device; `feature_extractor.resolve_features(target_stems, target_features)` does the same outside the model."""

    def __init__(self, data_path=None, separated_path="/nas/FMA/fma_separated/", use_preseparated=True,
                 scnet_separator=None, clip_duration=10.0, sample_rate=44100, n_fft=1024, hop_length=256, n_mels=128,
                 use_detailed_spectral=False, n_spectral_bins=32, compute_features="deferred", stem_loader=None,
                 stem_ext=".mp3", audio_loader=None, device=None):
        self.data_path, self.separated_path, self.use_preseparated = data_path, separated_path, use_preseparated
        self.scnet = scnet_separator
        self.clip_duration, self.sr = clip_duration, sample_rate
        self.n_fft, self.hop_length, self.n_mels = n_fft, hop_length, n_mels
        self.clip_samples = int(clip_duration * sample_rate)
        self.compute_features, self.device = compute_features, device
        self.stem_loader = stem_loader or default_stem_loader
        self.audio_loader = audio_loader or default_stem_loader
        self.stem_ext = stem_ext
        if use_preseparated:
            if not os.path.exists(separated_path):
                raise ValueError(f"Separated stems directory not found: {separated_path}")
            self.track_dirs = [d for d in glob.glob(os.path.join(separated_path, "*")) if os.path.isdir(d)]
        else:
            self.audio_files = []
            for ext in ("*.mp3", "*.wav", "*.flac"):
                self.audio_files.extend(glob.glob(os.path.join(data_path, "**", ext), recursive=True))
            if scnet_separator is None:
                raise ValueError("scnet_separator required when use_preseparated=False")
        self.feature_extractor = MixingFeatureExtractor(sample_rate, n_fft, hop_length, n_mels,
                                                        use_detailed_spectral=use_detailed_spectral,
                                                        n_spectral_bins=n_spectral_bins)

    def __len__(self):
        return len(self.track_dirs) if self.use_preseparated else len(self.audio_files)

    def _load_full(self, idx):
        if self.use_preseparated:
            stems = {}
            for name in STEMS:
                path = os.path.join(self.track_dirs[idx], f"{name}{self.stem_ext}")
                if not os.path.exists(path):
                    raise FileNotFoundError(f"Missing stem: {path}")
                audio, sr = self.stem_loader(path)
                audio = audio.float()
                if sr != self.sr:   # src/data.py:425-427
                    try:
                        import torchaudio
                        audio = torchaudio.transforms.Resample(sr, self.sr)(audio)
                    except ImportError:
                        audio = resample_sinc_hann(audio, sr, self.sr)
                stems[name] = audio.repeat(2, 1) if audio.shape[0] == 1 else audio
            return stems
        audio, sr = self.audio_loader(self.audio_files[idx])
        audio = audio.float()
        return self.scnet.separate(audio.repeat(2, 1) if audio.shape[0] == 1 else audio)

    def _random_crop_stems(self, stems_dict, duration_samples):
        total = next(iter(stems_dict.values())).shape[1]
        if total <= duration_samples:
            return {k: torch.nn.functional.pad(v, (0, duration_samples - total)) for k, v in stems_dict.items()}
        start = int(np.random.randint(0, total - duration_samples))
        return {k: v[:, start:start + duration_samples] for k, v in stems_dict.items()}

    def __getitem__(self, idx):
        if len(self) < 2:
            raise ValueError("StyleTransferDataset needs at least two songs (the target must differ from the input)")
        input_stems = self._random_crop_stems(self._load_full(idx), self.clip_samples)
        target_idx = int(np.random.randint(0, len(self)))
        while target_idx == idx:
            target_idx = int(np.random.randint(0, len(self)))
        target_stems = self._random_crop_stems(self._load_full(target_idx), self.clip_samples)
        return input_stems, target_stems, _clip_features(self, target_stems)


def style_transfer_collate_fn(batch):
    """[(input_stems, target_stems, target_features)] -> (input {stem: (B,2,T)}, target {stem: (B,2,T)}, (B,F))
    reference src/data.py:541-578."""
    inp = {s: torch.stack([b[0][s] for b in batch], 0) for s in STEMS}
    tgt = {s: torch.stack([b[1][s] for b in batch], 0) for s in STEMS}
    return inp, tgt, torch.stack([b[2] for b in batch], 0)
