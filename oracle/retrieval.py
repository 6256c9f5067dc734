"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of the reference's retrieval metric and of the
style-transfer sampling order -- plain loops, one query at a time, exactly as the reference walks them.

retrieve_top_k / evaluate_retrieval_accuracy: reference src/validation_utils.py:217-240, :243-282.
style_transfer_draws: the numpy global-RNG draw order of StyleTransferDataset.__getitem__, src/data.py:467-487 (crop:
`randint(0, total - duration)` only when total > duration) and :511-519 (target index redrawn while equal to idx).
Parity: pinned by construction on seeded inputs only (the reference has no fixtures for these; data.py cannot be imported
here because of its un-vendored SCNet imports -- SURVEY 8c) => "parity unpinned" for the sampling order."""
import numpy as np
import torch
import torch.nn.functional as F


def retrieve_top_k(query_embedding, retrieval_pool, k=5):
    q = F.normalize(query_embedding.unsqueeze(0), dim=1)
    p = F.normalize(retrieval_pool, dim=1)
    sims = torch.matmul(q, p.T).squeeze(0)
    s, i = torch.topk(sims, k=k, largest=True)
    return i, s


def evaluate_retrieval_accuracy(queries, retrieval_pool, query_indices, pool_indices, k_values=(1, 5)):
    correct = {k: 0 for k in k_values}
    for i in range(queries.shape[0]):
        top, _ = retrieve_top_k(queries[i], retrieval_pool, k=max(k_values))
        got = [pool_indices[j.item()] for j in top]
        for k in k_values:
            if query_indices[i] in got[:k]:
                correct[k] += 1
    return {f"top_{k}_accuracy": correct[k] / queries.shape[0] for k in k_values}


def style_transfer_draws(idx, lengths, clip_samples):
    """-> (input_start | None, target_idx, target_start | None); None = padded, no draw."""
    def crop(total):
        return None if total <= clip_samples else int(np.random.randint(0, total - clip_samples))
    a = crop(lengths[idx])
    t = int(np.random.randint(0, len(lengths)))
    while t == idx:
        t = int(np.random.randint(0, len(lengths)))
    return a, t, crop(lengths[t])
