"""Oracle: STFT power spectrogram + HTK mel filterbank + log-mel (CPU, fp32).

Follows torchaudio.transforms.MelSpectrogram as used at
reference src/mixing_utils.py:45-51,159 and src/model.py:33-39,41-67.
"""
import math

import torch

STEMS = ("vocals", "bass", "drums", "other")


def hann_periodic(n_fft: int) -> torch.Tensor:
    return torch.hann_window(n_fft, periodic=True, dtype=torch.float32)


def htk_fbank(sample_rate: int, n_fft: int, n_mels: int) -> torch.Tensor:
    """(n_fft//2+1, n_mels) triangular HTK filterbank, norm=None, f_max=sr//2.

    fp32 `linspace` arithmetic exactly as torchaudio.functional.melscale_fbanks.
    """
    n_freqs = n_fft // 2 + 1
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    m_pts = torch.linspace(hz2mel(0.0), hz2mel(float(sample_rate // 2)), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    width = f_pts[1:] - f_pts[:-1]
    d = f_pts[None, :] - freqs[:, None]
    falling = (-1.0 * d[:, :-2]) / width[:-1]
    rising = d[:, 2:] / width[1:]
    return torch.clamp(torch.minimum(falling, rising), min=0.0)


def power_spectrogram(x: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    """x (..., T) -> (..., n_fft//2+1, 1+T//hop); center/reflect, periodic Hann, |X|^2."""
    lead = x.shape[:-1]
    X = torch.stft(x.reshape(-1, x.shape[-1]), n_fft, hop, n_fft, hann_periodic(n_fft).to(x.dtype),
                   center=True, pad_mode="reflect", normalized=False, onesided=True,
                   return_complex=True)
    return X.abs().pow(2.0).reshape(lead + X.shape[-2:])


def mel_power(x: torch.Tensor, sample_rate=44100, n_fft=1024, hop=256, n_mels=128) -> torch.Tensor:
    """x (..., T) -> (..., n_mels, frames) mel power (reference `mel_transform(audio)`)."""
    spec = power_spectrogram(x, n_fft, hop)
    fb = htk_fbank(sample_rate, n_fft, n_mels).to(x.dtype)
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)


def logmel(stems: torch.Tensor, sample_rate=44100, n_fft=1024, hop=256, n_mels=128) -> torch.Tensor:
    """stems (B, 8, T), channel order vL,vR,bL,bR,dL,dR,oL,oR -> (B, 8, n_mels, frames).

    reference src/model.py:41-67 (MelSpectrogramPreprocessor.forward): natural log(mel + 1e-10).
    """
    return torch.log(mel_power(stems, sample_rate, n_fft, hop, n_mels) + 1e-10)


def stems_dict_to_tensor(stems_dict) -> torch.Tensor:
    """{stem: (..., 2, T)} -> (..., 8, T) in the reference channel order."""
    return torch.cat([stems_dict[s] for s in STEMS], dim=-2)


def tensor_to_stems_dict(x: torch.Tensor):
    return {s: x[..., 2 * i:2 * i + 2, :] for i, s in enumerate(STEMS)}
