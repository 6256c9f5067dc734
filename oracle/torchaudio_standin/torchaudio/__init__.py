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


def load(*a, **k):  # pragma: no cover - data.py cannot import here anyway (SCNet absent)
    raise RuntimeError("torchaudio stand-in: load() is not provided")
