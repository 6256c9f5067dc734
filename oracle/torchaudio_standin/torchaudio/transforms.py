"""Stand-in `torchaudio.transforms` (see package docstring).  Only MelSpectrogram."""
import math

import torch


def _hz_to_mel_htk(f: float) -> float:
    return 2595.0 * math.log10(1.0 + (f / 700.0))


def melscale_fbanks_htk(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int):
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel_htk(f_min), _hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


class Spectrogram(torch.nn.Module):
    def __init__(self, n_fft, hop_length, power):
        super().__init__()
        self.n_fft, self.hop_length, self.power = n_fft, hop_length, power
        self.register_buffer("window", torch.hann_window(n_fft))

    def forward(self, x):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        s = torch.stft(x2, self.n_fft, self.hop_length, self.n_fft, self.window, center=True,
                       pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        s = s.reshape(shape[:-1] + s.shape[-2:])
        return s.abs().pow(self.power)


class MelScale(torch.nn.Module):
    def __init__(self, n_mels, sample_rate, n_stft):
        super().__init__()
        fb = melscale_fbanks_htk(n_stft, 0.0, float(sample_rate // 2), n_mels, sample_rate)
        self.register_buffer("fb", fb)

    def forward(self, spec):
        return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)


class MelSpectrogram(torch.nn.Module):
    def __init__(self, sample_rate=16000, n_fft=400, hop_length=None, n_mels=128, power=2.0, **kw):
        super().__init__()
        hop_length = hop_length if hop_length is not None else n_fft // 2
        self.sample_rate, self.n_fft, self.hop_length, self.n_mels = sample_rate, n_fft, hop_length, n_mels
        self.spectrogram = Spectrogram(n_fft, hop_length, power)
        self.mel_scale = MelScale(n_mels, sample_rate, n_fft // 2 + 1)

    def forward(self, waveform):
        return self.mel_scale(self.spectrogram(waveform))


class Resample(torch.nn.Module):  # pragma: no cover
    def __init__(self, *a, **k):
        super().__init__()
        raise RuntimeError("torchaudio stand-in: Resample is not provided")
