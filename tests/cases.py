"""Shared seeded input builders for oracle / golden / GPU parity tests (data only, no reference code)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mst_amd.synth import synth_clip  # noqa: E402

STEMS = ("vocals", "bass", "drums", "other")

CFG_DEFAULT = dict(sample_rate=44100, n_fft=1024, hop_length=256, n_mels=128, split_size=20, overlap=10,
                   embed_dim=768)
CFG_BASELINE_SH = dict(sample_rate=44100, n_fft=2048, hop_length=512, n_mels=80, split_size=16, overlap=8,
                       embed_dim=512)  # reference scripts/train_baseline.sh:41-48


def _g(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def feature_case(name: str, T: int = 44100) -> torch.Tensor:
    """(8, T) fp32 stems for a named edge case."""
    if name == "synth":
        return synth_clip(0, T)
    if name == "synth1":
        return synth_clip(1, T)
    if name == "white":          # SURVEY A.5 sanity vector
        return 0.1 * torch.randn(8, T, generator=_g(7))
    if name == "silent_vocals":  # crest = ILD = -100, corr = 0, tilt = 0, flatness = 1
        x = 0.1 * torch.randn(8, T, generator=_g(8))
        x[0:2] = 0.0
        return x
    if name == "mono":           # every stem L == R: MSR = 0, corr ~ 1
        x = 0.1 * torch.randn(8, T, generator=_g(9))
        x[1::2] = x[0::2]
        return x
    if name == "dc":             # DC offsets exercise the centred correlation
        x = 0.05 * torch.randn(8, T, generator=_g(10))
        return x + torch.tensor([0.3, -0.2, 0.1, 0.1, 0.0, 0.5, -0.4, 0.25])[:, None]
    if name == "clipped":        # full-scale saturation
        return (1.5 * torch.randn(8, T, generator=_g(11))).clamp_(-1.0, 1.0)
    if name == "short_padded":   # clip shorter than clip_samples, zero-padded tail (data.py:283-287)
        x = synth_clip(3, T)
        x[:, int(0.4 * T):] = 0.0
        return x
    if name == "all_silent":
        return torch.zeros(8, T)
    if name == "one_sided":      # R silent in every stem: ILD=+100 clamp, MSR = 1
        x = 0.1 * torch.randn(8, T, generator=_g(12))
        x[1::2] = 0.0
        return x
    raise KeyError(name)


FEATURE_CASES = ("synth", "white", "silent_vocals", "mono", "dc", "clipped", "short_padded",
                 "all_silent", "one_sided")


def n_subbands(n_mels, split_size, overlap):
    return len(range(0, n_mels - split_size + 1, overlap))


def state_dict_shapes(cfg, feature_dim=64):
    ns = n_subbands(cfg["n_mels"], cfg["split_size"], cfg["overlap"])
    sub = max(1, cfg["split_size"] // 10)
    freq_dim = (cfg["split_size"] // sub) // 4
    C = 64 * ns * freq_dim
    E = cfg["embed_dim"]
    shapes = {}
    for i in range(ns):
        p = f"audio_encoder.subnet_cnns.{i}."
        for conv, bn, co, ci in (("conv1", "bn1", 32, 8), ("conv2", "bn2", 64, 32)):
            shapes[p + conv + ".weight"] = (co, ci, 7, 7)
            shapes[p + conv + ".bias"] = (co,)
            for k in ("weight", "bias", "running_mean", "running_var"):
                shapes[p + bn + "." + k] = (co,)
            shapes[p + bn + ".num_batches_tracked"] = ()
    a = "audio_encoder.attention_pooling."
    shapes[a + "attention.0.weight"] = (256, C)
    shapes[a + "attention.0.bias"] = (256,)
    shapes[a + "attention.2.weight"] = (1, 256)
    shapes[a + "attention.2.bias"] = (1,)
    shapes[a + "projection.0.weight"] = (E, C)
    shapes[a + "projection.0.bias"] = (E,)
    f = "film_encoder."
    shapes[f + "feature_mlp.0.weight"] = (256, feature_dim)
    shapes[f + "feature_mlp.0.bias"] = (256,)
    shapes[f + "feature_mlp.3.weight"] = (256, 256)
    shapes[f + "feature_mlp.3.bias"] = (256,)
    shapes[f + "film_head.weight"] = (ns * 192, 256)
    shapes[f + "film_head.bias"] = (ns * 192,)
    return shapes


def make_state_dict(cfg, seed=42, feature_dim=64):
    """Deterministic reference-format state_dict (trained-looking: non-trivial BN stats, FiLM gamma ~ 1)."""
    g = _g(seed)
    sd = {}
    for k, shp in state_dict_shapes(cfg, feature_dim).items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(100, dtype=torch.long)
        elif k.endswith("running_var") or (".bn" in k and k.endswith(".weight")):
            sd[k] = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("running_mean") or (".bn" in k and k.endswith(".bias")):
            sd[k] = 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".weight"):
            fan_in = math.prod(shp[1:])
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan_in)
        else:
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
    # FiLM gammas around 1 so activations stay O(1) through both conv blocks
    ns = n_subbands(cfg["n_mels"], cfg["split_size"], cfg["overlap"])
    b = sd["film_encoder.film_head.bias"]
    for i in range(ns):
        b[i * 192:i * 192 + 32] += 1.0
        b[i * 192 + 64:i * 192 + 128] += 1.0
    return sd


def pcm_clip(c: int, T: int) -> torch.Tensor:
    """(8, T) int16 PCM stems built with INTEGER arithmetic only (torch.randint + shifts / integer sums): bit-identical on
    every machine, unlike `synth_clip`, whose sin / exp / randn differ in the last bit between CPU generations (1e-7
    relative -- harmless under the 1e-4 forward tolerances, but enough to flip a max-pool arg-max in a gradient fixture).
    vocals: noise, silent first fifth; bass: mono moving sum over 64 samples (low-pass); drums: noise under a sawtooth decay of
    4096 samples; other: independent L / R, R at 45/64.  Value = sample / 32768."""
    g = _g(7000 + c)
    n = torch.randint(-2048, 2048, (8, T), generator=g, dtype=torch.int32)
    v = n[0:2].clone()
    v[:, :T // 5] = 0
    cs = torch.cumsum(n[2].to(torch.int64), 0)
    ma = cs.clone()
    ma[64:] -= cs[:-64]
    b = (ma >> 3).to(torch.int32)
    env = (4096 - (torch.arange(T) % 4096)).to(torch.int32)
    d = (n[4:6] * env) >> 12
    o = torch.stack([n[6], (n[7] * 45) >> 6])
    return torch.cat([v, torch.stack([b, b]), d, o]).clamp(-32768, 32767).to(torch.int16)


def pcm_batch(B: int, T: int) -> torch.Tensor:
    """(B, 8, T) fp32 = pcm_clip / 32768 (exact)."""
    return torch.stack([pcm_clip(c, T) for c in range(B)], 0).float() / 32768.0


def sample_idx(n, k=2048, seed=0):
    """Deterministic sample of k of n flat indices (the goldens store the values only, not the indices)."""
    import numpy as np
    return np.random.default_rng(seed).choice(n, size=min(k, n), replace=False).astype(np.int64)


def checksum(x: torch.Tensor):
    x = x.double()
    return [float(x.sum()), float((x * x).sum())]


def pseudo_separate(x2: torch.Tensor, sample_rate: int = 44100) -> torch.Tensor:
    """Fixed deterministic 4-way split of a stereo mix (2, T) -> stems (8, T), used where the reference would call
    SCNet (absent: BASELINE.md section 4).  bass = LP 200 Hz, drums = HP 4 kHz, vocals = 0.6 * mid of the rest
    (mono), other = the remainder; the stems sum back to the mix.  scipy float64 filters, cast to fp32."""
    import numpy as np
    from scipy.signal import butter, sosfilt
    x = x2.double().numpy()
    bass = sosfilt(butter(2, 200, btype="low", fs=sample_rate, output="sos"), x, axis=-1)
    drums = sosfilt(butter(2, 4000, btype="high", fs=sample_rate, output="sos"), x, axis=-1)
    rest = x - bass - drums
    mid = 0.6 * rest.mean(axis=0, keepdims=True).repeat(2, axis=0)
    other = rest - mid
    stems = np.concatenate([mid, bass, drums, other], axis=0)   # vocals, bass, drums, other
    return torch.from_numpy(stems.astype(np.float32))


def song_a_clips():
    """The two 10 s crops of the reference's assets/song_A.wav (samples 0 and 220538; BASELINE configs[0]) from the
    committed int16 fixture, pseudo-separated: (2, 8, 441000) fp32."""
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "song_a_crops.npz"))
    out = []
    for k in ("crop0", "crop1"):
        x = torch.from_numpy(g[k].astype(np.float32) / 32768.0)   # (2, 441000)
        out.append(pseudo_separate(x))
    return torch.stack(out, 0)


# ---------------------------------------------------------------------------------------------------------------
# Toy pre-separated track directories for the dataset / retrieval fixtures (tests/golden/dataset.npz)
# ---------------------------------------------------------------------------------------------------------------
TOY_CLIP = 11025            # 0.25 s at 44.1 kHz
TOY_TRACKS = {              # name -> (length in samples, channels per stem file); lengths vs C=11025: >2C, ==2C, <2C, <C
    "a_long": (30000, 2), "b_exact2c": (22050, 2), "c_lt2c": (15000, 2), "d_ltc": (8000, 2), "e_mono": (26000, 1)}


def toy_track_pcm(name):
    """int16 (8, L) stems of a toy track (mono tracks: R == L)."""
    L, ch = TOY_TRACKS[name]
    x = synth_clip(100 + sorted(TOY_TRACKS).index(name), L)
    q = torch.round(x * 32767.0).to(torch.int16)
    if ch == 1:
        q[1::2] = q[0::2]
    return q


def write_toy_tracks(root, ext=".mp3"):
    """One directory per toy track holding {stem}{ext} as 16-bit PCM RIFF bytes (the reference hard-codes the `.mp3`
    names, src/data.py:188; the fixture generator reads them through a RIFF-reading stand-in)."""
    import wave
    for name, (L, ch) in TOY_TRACKS.items():
        d = os.path.join(root, name)
        os.makedirs(d, exist_ok=True)
        q = toy_track_pcm(name)
        for i, s in enumerate(STEMS):
            a = q[2 * i:2 * i + ch].T.contiguous().numpy().astype("<i2")
            with wave.open(os.path.join(d, s + ext), "wb") as w:
                w.setnchannels(ch)
                w.setsampwidth(2)
                w.setframerate(44100)
                w.writeframes(a.tobytes())
    return root


class RandintLog:
    """Context manager that records every np.random.randint(low, high) call made inside it as (low, high, result)."""

    def __enter__(self):
        import numpy as np
        self.np, self.real, self.calls = np, np.random.randint, []

        def logged(low, high=None, *a, **k):
            v = self.real(low, high, *a, **k)
            self.calls.append((int(low), -1 if high is None else int(high), int(v)))
            return v
        np.random.randint = logged
        return self

    def __exit__(self, *exc):
        self.np.random.randint = self.real
        return False
