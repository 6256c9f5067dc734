"""Retrieval validation on the HIP path -- mirror of reference src/validation_utils.py.

Same function names, arguments and return values.  What changes underneath:
  * `compute_embedding` / `compute_track_embedding` run features + log-mel + encoder in libmst.so (one pass over the
    waveform instead of the reference's feature pass + model pass, validation_utils.py:140-146);
  * `build_embedding_cache` batches tracks (`batch_size`) instead of one forward per track (:186-205);
  * `evaluate_retrieval_accuracy` is one similarity matmul + top-k for all queries instead of a Python loop (:262-276);
  * audio decoding: the reference uses librosa (not installed here); segments are read through the pluggable
    `stem_loader` of mst_amd.data (torchaudio if present, PCM .wav otherwise) or straight from `.pcm16` shards.
"""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import ingest
from .data import default_stem_loader
from .mixing_utils import STEMS


def _fit(audio: torch.Tensor, start, n):
    seg = audio[:, start:start + n]
    if seg.shape[0] == 1:
        seg = seg.repeat(2, 1)
    if seg.shape[1] < n:
        seg = F.pad(seg, (0, n - seg.shape[1]))
    return seg[:2]


def load_audio_segment(audio_path, start_sec, duration_sec, sample_rate=44100, loader=None):
    """-> np.ndarray (2, samples) stereo, zero-padded if short (reference :15-46)."""
    audio, sr = (loader or default_stem_loader)(audio_path)
    if sr != sample_rate:
        raise RuntimeError(f"{audio_path}: sample rate {sr} != {sample_rate} (resampling needs torchaudio)")
    return _fit(audio.float(), int(start_sec * sample_rate), int(duration_sec * sample_rate)).numpy()


def load_stems_segment(track_dir, start_sec, duration_sec, sample_rate=44100, loader=None, stem_ext=".mp3"):
    """-> {stem: np.ndarray (2, samples)} from a pre-separated track directory or a `.pcm16` shard (reference :49-74)."""
    s0, n = int(start_sec * sample_rate), int(duration_sec * sample_rate)
    if os.path.isfile(track_dir) and track_dir.endswith(".pcm16"):
        mm, sr = ingest.open_pcm_shard(track_dir)
        if sr != sample_rate:
            raise RuntimeError(f"{track_dir}: sample rate {sr} != {sample_rate}")
        seg = np.zeros((8, n), dtype=np.float32)
        k = max(0, min(n, mm.shape[1] - s0))
        seg[:, :k] = mm[:, s0:s0 + k].astype(np.float32) / 32768.0
        return {s: seg[2 * i:2 * i + 2] for i, s in enumerate(STEMS)}
    out = {}
    for name in STEMS:
        path = os.path.join(track_dir, f"{name}{stem_ext}")
        if not os.path.exists(path):
            raise FileNotFoundError(f"Stem file not found: {path}")
        out[name] = load_audio_segment(path, start_sec, duration_sec, sample_rate, loader)
    return out


def _to_dev(stems_dict, device):
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v).float().to(device) for k, v in stems_dict.items()}


def compute_embedding(stems_dict, mixing_features, model, device):
    """stems (numpy (2,T) each) + features (F,) -> embedding (E,) on the CPU (reference :77-103)."""
    st = {k: v.unsqueeze(0) for k, v in _to_dev(stems_dict, device).items()}
    with torch.no_grad():
        emb = model(st, mixing_features.unsqueeze(0).to(device))
    return emb.squeeze(0).cpu()


def compute_batch_embeddings(stems_batch, model, feature_extractor, device):
    """{stem: (B,2,T)} -> (B,E) on the device: one stage-A launch (features + log-mel) and one encoder forward."""
    st = _to_dev(stems_batch, device)
    with torch.no_grad():
        feats, logmel = feature_extractor.features_and_logmel(st)
        return model.forward_from_logmel(logmel, feats)


def compute_track_embedding(track_path, start_sec, duration_sec, model, feature_extractor, scnet, device,
                            use_preseparated=True, loader=None, stem_ext=".mp3"):
    """Embedding (E,) of one track segment (reference :106-148)."""
    if use_preseparated:
        stems = load_stems_segment(track_path, start_sec, duration_sec, 44100, loader, stem_ext)
    else:
        audio = load_audio_segment(track_path, start_sec, duration_sec, 44100, loader)
        stems = {k: v.cpu().numpy() for k, v in scnet.separate(torch.from_numpy(audio).float().to(device)).items()}
    batch = {k: torch.from_numpy(np.ascontiguousarray(v))[None] for k, v in stems.items()}
    return compute_batch_embeddings(batch, model, feature_extractor, device)[0].cpu()


def build_embedding_cache(dataset, indices, model, feature_extractor, scnet, device, query_duration=1.0,
                          use_preseparated=True, desc="Building cache", batch_size=16, loader=None, stem_ext=".mp3"):
    """{'embeddings': (N,E) CPU, 'track_indices': [...], 'track_paths': [...]} for the first `query_duration` seconds of
    every track; tracks that fail to load are reported and skipped (reference :151-214)."""
    embs, track_indices, track_paths = [], [], []
    pend_stems, pend_meta = [], []

    def flush():
        if not pend_stems:
            return
        batch = {s: torch.stack([torch.from_numpy(np.ascontiguousarray(p[s])) for p in pend_stems], 0) for s in STEMS}
        e = compute_batch_embeddings(batch, model, feature_extractor, device).cpu()
        for (idx, path), row in zip(pend_meta, e):
            embs.append(row), track_indices.append(idx), track_paths.append(path)
        pend_stems.clear(), pend_meta.clear()

    for idx in indices:
        try:
            path = dataset.track_dirs[idx] if use_preseparated else dataset.audio_files[idx]
            if use_preseparated:
                stems = load_stems_segment(path, 0.0, query_duration, 44100, loader, stem_ext)
            else:
                audio = load_audio_segment(path, 0.0, query_duration, 44100, loader)
                stems = {k: v.cpu().numpy() for k, v in scnet.separate(torch.from_numpy(audio).float().to(device)).items()}
            pend_stems.append(stems), pend_meta.append((idx, path))
            if len(pend_stems) == batch_size:
                flush()
        except Exception as e:  # reference behaviour: report, continue
            print(f"\nError processing track {idx}: {e}")
    flush()
    return {"embeddings": torch.stack(embs) if embs else torch.empty(0, 0), "track_indices": track_indices,
            "track_paths": track_paths}


def retrieve_top_k(query_embedding, retrieval_pool, k=5):
    """(indices (k,), cosine similarities (k,)) of the k nearest pool rows (reference :217-240)."""
    sims = (F.normalize(query_embedding.unsqueeze(0), dim=1) @ F.normalize(retrieval_pool, dim=1).T).squeeze(0)
    top_s, top_i = torch.topk(sims, k=k, largest=True)
    return top_i, top_s


def evaluate_retrieval_accuracy(queries, retrieval_pool, query_indices, pool_indices, k_values=[1, 5]):
    """{'top_k_accuracy': fraction of queries whose own track is among the k nearest pool rows} (reference :243-282)."""
    kmax = max(k_values)
    sims = F.normalize(queries, dim=1) @ F.normalize(retrieval_pool, dim=1).T            # (M, N)
    top = torch.topk(sims, k=kmax, dim=1, largest=True).indices.cpu()                      # (M, kmax)
    pool = torch.as_tensor(pool_indices)[top]                                              # track ids
    hit = pool == torch.as_tensor(query_indices)[:, None]
    return {f"top_{k}_accuracy": hit[:, :k].any(dim=1).float().sum().item() / queries.shape[0] for k in k_values}


def save_cache(cache, save_path):
    os.makedirs(os.path.dirname(save_path), exist_ok=True)
    torch.save(cache, save_path)
    print(f"Cache saved to {save_path}")


def load_cache(cache_path):
    cache = torch.load(cache_path, map_location="cpu")
    print(f"Cache loaded from {cache_path}")
    return cache


def save_metrics(metrics, save_path):
    os.makedirs(os.path.dirname(save_path), exist_ok=True)
    with open(save_path, "w") as f:
        json.dump(metrics, f, indent=2)
    print(f"Metrics saved to {save_path}")
