"""Oracle: 64-d (or detailed-mode) mixing feature vector (CPU, fp32).

Restates reference src/mixing_utils.py:71-357 (MixingFeatureExtractor) batched over
clips.  Input stems (B, 8, T) in channel order vL,vR,bL,bR,dL,dR,oL,oR.
Output layout = the reference's sorted-key flattening (mixing_utils.py:320-357):
  [0:15] bass  = dynamics(6) rel_loudness(1) spectral(5) stereo(3)
  [15:30] drums, [30:34] masking(vocals,bass,drums,other), [34:49] other, [49:64] vocals
"""
import torch

from . import mel as omel

STEM_INDEX = {"vocals": 0, "bass": 1, "drums": 2, "other": 3}


def loudness(x: torch.Tensor) -> torch.Tensor:
    """mixing_utils.py:311-318.  x (B, C, T) -> (B,)"""
    rms = torch.sqrt(torch.mean(x ** 2, dim=(-2, -1)))
    return -0.691 + 10 * torch.log10(rms ** 2 + 1e-10)


def dynamics(x: torch.Tensor) -> torch.Tensor:
    """mixing_utils.py:107-139.  x (B, 2, T) -> (B, 6) [rmsL rmsR crestL crestR loud loud]"""
    rms = torch.sqrt(torch.mean(x ** 2, dim=-1))
    peak = x.abs().amax(dim=-1)
    crest = 20 * torch.log10(peak / (rms + 1e-8))
    ld = loudness(x)[:, None]
    return torch.cat([rms, crest, ld, ld], dim=1)


def _pearson_vs_index(y: torch.Tensor) -> torch.Tensor:
    """corrcoef(arange(n), y)[0,1] per row, 0 where unbiased std(y) < 1e-6 (mixing_utils.py:181-187)."""
    n = y.shape[-1]
    idx = torch.arange(n, dtype=torch.float32)
    out = torch.zeros(y.shape[0])
    for b in range(y.shape[0]):
        if y[b].std() < 1e-6:
            continue
        out[b] = torch.corrcoef(torch.stack([idx, y[b]]))[0, 1]
    return out


def spectral(x: torch.Tensor, melp: torch.Tensor, detailed=False, n_bins=32) -> torch.Tensor:
    """mixing_utils.py:141-236.  melp = mel power (B, 2, M, F) of this stem -> (B, 5) or (B, n_bins+2)."""
    M = melp.shape[-2]
    band_db = (10 * torch.log10(melp + 1e-10)).mean(dim=(1, 3))  # (B, M)
    flat = torch.exp(torch.log(melp + 1e-10).mean(dim=(1, 2, 3))) / (melp.mean(dim=(1, 2, 3)) + 1e-10)
    if not detailed:
        q = M // 4
        lo, mid, hi = band_db[:, :q].mean(1), band_db[:, q:3 * q].mean(1), band_db[:, 3 * q:].mean(1)
        tilt = _pearson_vs_index(band_db)
        return torch.stack([lo, mid, hi, tilt, flat], dim=1)
    if n_bins >= M:
        curve = band_db
    else:
        curve = torch.nn.functional.interpolate(band_db[:, None, :], size=n_bins, mode="linear",
                                                align_corners=True)[:, 0, :]
    # reference quirk (mixing_utils.py:220): freq axis is arange(n_spectral_bins) even if the
    # curve kept all n_mels points; that case raises in the reference, so it does here too.
    if curve.shape[-1] != n_bins:
        raise RuntimeError("detailed mode with n_spectral_bins >= n_mels is broken in the reference")
    tilt = _pearson_vs_index(curve)
    return torch.cat([curve, tilt[:, None], flat[:, None]], dim=1)


def stereo(x: torch.Tensor) -> torch.Tensor:
    """mixing_utils.py:238-268.  x (B, 2, T) -> (B, 3) [ILD, correlation, mid/side ratio]"""
    L, R = x[:, 0], x[:, 1]
    rl, rr = torch.sqrt((L ** 2).mean(-1)), torch.sqrt((R ** 2).mean(-1))
    ild = 20 * torch.log10(rl / (rr + 1e-8))
    Lc, Rc = L - L.mean(-1, keepdim=True), R - R.mean(-1, keepdim=True)
    corr = (Lc * Rc).sum(-1) / (torch.sqrt((Lc ** 2).sum(-1) * (Rc ** 2).sum(-1)) + 1e-8)
    e_mid, e_side = (((L + R) / 2) ** 2).mean(-1), (((L - R) / 2) ** 2).mean(-1)
    return torch.stack([ild, corr, e_side / (e_mid + 1e-8)], dim=1)


def masking(melp8: torch.Tensor) -> torch.Tensor:
    """mixing_utils.py:270-309.  melp8 (B, 8, M, F) mel power -> (B, 4) in stem order v,b,d,o."""
    S = melp8.reshape(melp8.shape[0], 4, 2, *melp8.shape[2:]).mean(dim=2)  # (B,4,M,F)
    out = []
    for i in range(4):
        others = torch.stack([S[:, j] for j in range(4) if j != i]).amax(dim=0)
        out.append(torch.sigmoid((0.0 - (S[:, i] - others)) / 1.0).mean(dim=(1, 2)))
    return torch.stack(out, dim=1)


def extract_all_features(stems: torch.Tensor, sample_rate=44100, n_fft=1024, hop=256, n_mels=128,
                         detailed=False, n_bins=32, return_mel=False):
    """stems (B, 8, T) fp32 -> (B, feature_dim).  mixing_utils.py:71-105 + :320-357."""
    stems = stems.float()
    B = stems.shape[0]
    melp = omel.mel_power(stems, sample_rate, n_fft, hop, n_mels)  # (B,8,M,F)
    s4 = stems.reshape(B, 4, 2, -1)
    mix = ((s4[:, 0] + s4[:, 1]) + s4[:, 2]) + s4[:, 3]  # python sum() over dict order v,b,d,o
    mix_l = loudness(mix)
    per = {}
    for name, i in STEM_INDEX.items():
        x = stems[:, 2 * i:2 * i + 2]
        per[name] = torch.cat([
            dynamics(x),
            (loudness(x) - mix_l)[:, None],
            spectral(x, melp[:, 2 * i:2 * i + 2], detailed, n_bins),
            stereo(x)], dim=1)
    vec = torch.cat([per["bass"], per["drums"], masking(melp), per["other"], per["vocals"]], dim=1)
    vec = torch.clamp(vec, min=-100.0, max=100.0)
    vec = torch.where(torch.isnan(vec), torch.zeros_like(vec), vec)
    if torch.isinf(vec).any():
        raise ValueError("Inf detected in features after clamping")
    return (vec, melp) if return_mel else vec
