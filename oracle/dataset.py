"""Oracle: crop-index and collate logic of FMABaselineDataset (reference src/data.py:201-328).

PARITY UNPINNED: src/data.py cannot be imported here (it imports the un-vendored SCNet
submodule at module scope); this is a restatement read from the source.
"""
import numpy as np
import torch

STEMS = ("vocals", "bass", "drums", "other")


def crop_starts(audio_length: int, clip_samples: int, num_segments: int):
    """Consumes the global numpy RNG exactly as data.py:222-266 does."""
    if num_segments == 1:
        m = audio_length - clip_samples
        return [0 if m <= 0 else int(np.random.randint(0, m + 1))]
    if num_segments == 2:
        if audio_length < 2 * clip_samples:
            return [0, 0]
        s1 = int(np.random.randint(0, audio_length - 2 * clip_samples + 1))
        s2 = int(np.random.randint(s1 + clip_samples, audio_length - clip_samples + 1))
        return [s1, s2]
    raise ValueError(f"num_segments={num_segments} is not supported. "
                     f"Only num_segments=1 or num_segments=2 are implemented.")


def extract_clip(stems_full, start: int, clip_samples: int):
    """data.py:276-288: slice, zero-pad the tail when short."""
    out = {}
    for k, a in stems_full.items():
        seg = a[:, start:start + clip_samples]
        if seg.shape[1] < clip_samples:
            seg = torch.nn.functional.pad(seg, (0, clip_samples - seg.shape[1]))
        out[k] = seg
    return out


def collate(batch):
    """data.py:291-328"""
    stems, feats, labels, dirs = [], [], [], []
    for stems_list, features_list, idx, track_dir in batch:
        for s, f in zip(stems_list, features_list):
            stems.append(s), feats.append(f), labels.append(idx), dirs.append(track_dir)
    sd = {k: torch.stack([s[k] for s in stems], 0) for k in STEMS}
    return sd, torch.stack(feats, 0), torch.tensor(labels, dtype=torch.long), dirs
