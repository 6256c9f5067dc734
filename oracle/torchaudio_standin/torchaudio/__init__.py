"""Minimal stand-in for the one torchaudio symbol the reference hot path uses.

TEST INFRASTRUCTURE ONLY.  torchaudio is not installed in the build container
(SURVEY.md F6).  This module restates torchaudio's *documented* algorithm for
`transforms.MelSpectrogram` (Spectrogram: periodic Hann, torch.stft center/reflect,
|X|^power; MelScale: HTK triangular filterbank, norm=None) so that the reference's
own `src/mixing_utils.py` / `src/model.py` can be imported from /root/reference in
this container to generate golden vectors (tests/golden/make_golden.py).

It is never imported by the product package and never travels as a dependency of it.
Equality with a real torchaudio install could not be verified offline; parity at the
mel boundary is therefore "pinned to algorithm" (DESIGN.md section Oracle).
"""
from . import transforms  # noqa: F401

__version__ = "0.0-standin"


def load(path, *a, **k):
    """16-bit PCM RIFF reader -> (float32 tensor (channels, frames) scaled by 1/32768, sample_rate), what
    torchaudio.load(normalize=True) returns for such a file.  Used only to let the reference's src/data.py read the
    toy stems of tests/golden/make_golden.py::gen_dataset (RIFF bytes under the hard-coded `{stem}.mp3` names)."""
    import wave

    import numpy as np
    import torch
    with wave.open(str(path), "rb") as w:
        ch, sw, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        if sw != 2:
            raise RuntimeError("torchaudio stand-in: 16-bit PCM only")
        raw = w.readframes(n)
    a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    return torch.from_numpy(a.reshape(-1, ch).T.copy()), sr
