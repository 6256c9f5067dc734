"""Oracle: InfoNCE (reference src/loss.py:31-136), vectorised, CPU fp32."""
import torch
import torch.nn.functional as F


def info_nce(emb: torch.Tensor, labels: torch.Tensor, temperature=0.1) -> torch.Tensor:
    e = F.normalize(emb, dim=1)
    sim = e @ e.T / temperature
    lab = labels[:, None]
    same = (lab == lab.T)
    eye = torch.eye(len(labels), dtype=torch.bool)
    pos_m = (same & ~eye).float()
    neg_m = (~same).float()
    ex = torch.exp(sim - sim.max(dim=1, keepdim=True)[0])
    pos, neg = (ex * pos_m).sum(1), (ex * neg_m).sum(1)
    keep = pos > 0
    if not keep.any():
        raise RuntimeError("No positive pairs found in batch!")
    return (-torch.log(pos[keep] / (pos[keep] + neg[keep] + 1e-8))).mean()
