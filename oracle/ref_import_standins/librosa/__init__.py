"""Stand-in `librosa` for importing the reference's data.py / validation_utils.py (see ../README.md).
TEST INFRASTRUCTURE ONLY.  Only `load` on 16-bit PCM RIFF files, no resampling."""
import wave

import numpy as np


def load(path, sr=22050, mono=True, offset=0.0, duration=None, **kw):
    with wave.open(str(path), "rb") as w:
        ch, sw, native, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        if sw != 2:
            raise RuntimeError("librosa stand-in: 16-bit PCM only")
        start = min(int(offset * native), n)
        w.setpos(start)
        count = n - start if duration is None else min(int(duration * native), n - start)
        raw = w.readframes(count)
    y = (np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0).reshape(-1, ch).T
    if sr is not None and sr != native:
        raise RuntimeError("librosa stand-in: resampling is not provided")
    if mono or ch == 1:
        y = y.mean(axis=0) if ch > 1 else y[0]
    return np.ascontiguousarray(y), native
