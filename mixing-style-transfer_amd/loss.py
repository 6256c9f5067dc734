"""InfoNCE loss with the reference semantics (src/loss.py:10-136), vectorised (no per-anchor host syncs) and
sharded: with torch.distributed initialised, embeddings and labels are all-gathered over RCCL/xGMI so every
rank scores its local anchors against the global batch."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _AllGatherWithGrad(torch.autograd.Function):
    """all-gather whose backward all-reduces the gradient of the gathered tensor and returns the local slice, so the
    gradient w.r.t. the local embeddings includes the terms where they act as *columns* of other ranks' anchors."""

    @staticmethod
    def forward(ctx, x):
        import torch.distributed as dist
        ws, ctx.rank, ctx.n = dist.get_world_size(), dist.get_rank(), x.shape[0]
        out = torch.empty(ws * ctx.n, *x.shape[1:], dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous())
        return out

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        g = g.contiguous().clone()
        dist.all_reduce(g)
        return g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n]


def gather_embeddings(emb: torch.Tensor, labels: torch.Tensor):
    """All-gather (N_local, D) fp32 embeddings + (N_local,) int64 labels over RCCL/xGMI -> global tensors in rank
    order, plus the row offset of the local slice.  Differentiable w.r.t. `emb` (see _AllGatherWithGrad)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return emb, labels, 0
    ws, rank = dist.get_world_size(), dist.get_rank()
    n = emb.shape[0]
    all_l = torch.empty(ws * n, dtype=labels.dtype, device=labels.device)
    dist.all_gather_into_tensor(all_l, labels.contiguous())
    if emb.requires_grad:
        all_e = _AllGatherWithGrad.apply(emb)
    else:
        all_e = torch.empty(ws * n, emb.shape[1], dtype=emb.dtype, device=emb.device)
        dist.all_gather_into_tensor(all_e, emb.contiguous())
    return all_e, all_l, rank * n


def info_nce_rows(all_emb, all_labels, row0, rows, temperature):
    """Sum of -log(pos/(pos+neg+1e-8)) over local anchors [row0,row0+rows) that have a positive, and their count."""
    e = F.normalize(all_emb, dim=1)
    sim = e[row0:row0 + rows] @ e.T / temperature            # (rows, N)
    lab = all_labels
    same = lab[row0:row0 + rows, None] == lab[None, :]
    self_mask = torch.zeros_like(same)
    self_mask[torch.arange(rows, device=sim.device), torch.arange(row0, row0 + rows, device=sim.device)] = True
    ex = torch.exp(sim - sim.max(dim=1, keepdim=True)[0])
    pos = (ex * (same & ~self_mask)).sum(1)
    neg = (ex * (~same)).sum(1)
    keep = pos > 0
    li = -torch.log(pos / (pos + neg + 1e-8))
    return torch.where(keep, li, torch.zeros_like(li)).sum(), keep.sum()


def _infonce_ws(N, D, device):
    from . import _lib
    need = _lib.lib().mst_infonce_workspace_bytes(N, D)
    return torch.empty(need, dtype=torch.uint8, device=device), need


class _InfoNCERowsHip(torch.autograd.Function):
    """(sum of row losses, #valid rows) in libmst.so; backward = `mst_infonce_backward` (gradient for all N rows)."""

    @staticmethod
    def forward(ctx, all_emb, all_labels, row0, rows, temperature):
        from . import _lib
        e = all_emb.detach().contiguous().float()
        lab = all_labels.contiguous().to(torch.int64)
        N, D = e.shape
        ws, need = _infonce_ws(N, D, e.device)
        out = torch.empty(2, dtype=torch.float32, device=e.device)
        with torch.cuda.device(e.device):
            _lib.check(_lib.lib().mst_infonce_forward(_lib.dptr(e), _lib.dptr(lab), N, D, row0, rows, float(temperature),
                                                      _lib.dptr(out), _lib.dptr(ws), need, _lib.stream_ptr(e.device)),
                       "mst_infonce_forward")
        ctx.save_for_backward(e, lab)
        ctx.cfg = (row0, rows, float(temperature))
        ctx.mark_non_differentiable(out[1])
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_sum, _g_cnt):
        from . import _lib
        e, lab = ctx.saved_tensors
        row0, rows, temperature = ctx.cfg
        N, D = e.shape
        ws, need = _infonce_ws(N, D, e.device)
        grad = torch.empty_like(e)
        scale = g_sum.detach().reshape(1).float().contiguous()
        with torch.cuda.device(e.device):
            _lib.check(_lib.lib().mst_infonce_backward(_lib.dptr(e), _lib.dptr(lab), N, D, row0, rows, temperature,
                                                       _lib.dptr(scale), _lib.dptr(grad), _lib.dptr(ws), need,
                                                       _lib.stream_ptr(e.device)), "mst_infonce_backward")
        return grad, None, None, None, None


def info_nce_rows_hip(all_emb, all_labels, row0, rows, temperature):
    """Same as info_nce_rows, in libmst.so (`mst_infonce_forward` / `mst_infonce_backward`); differentiable."""
    return _InfoNCERowsHip.apply(all_emb, all_labels, row0, rows, temperature)


class InfoNCELoss(nn.Module):
    """Drop-in for reference InfoNCELoss(temperature)(embeddings (N, D), song_labels (N,)) -> scalar."""

    def __init__(self, temperature=0.1, gather=False):
        super().__init__()
        self.temperature = temperature
        self.gather = gather

    def forward(self, embeddings, song_labels):
        if self.gather:
            import torch.distributed as dist
            all_e, all_l, row0 = gather_embeddings(embeddings, song_labels)
            rows_fn = info_nce_rows_hip if all_e.is_cuda else info_nce_rows
            s, c = rows_fn(all_e, all_l, row0, embeddings.shape[0], self.temperature)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                sc = torch.stack([s.detach(), c.to(s.dtype)])
                dist.all_reduce(sc)
                if sc[1].item() == 0:
                    raise RuntimeError("No positive pairs found in batch!")
                # value: the global mean over all valid anchors (identical on every rank); gradient: this rank's share
                # s / C_global -- summed over ranks (all-reduce SUM of parameter grads) it is the exact global gradient.
                share = s / sc[1]
                return share + (sc[0] / sc[1] - share).detach() if s.requires_grad else sc[0] / sc[1]
        else:
            rows_fn = info_nce_rows_hip if embeddings.is_cuda else info_nce_rows
            s, c = rows_fn(embeddings, song_labels, 0, embeddings.shape[0], self.temperature)
        if c.item() == 0:
            raise RuntimeError(
                f"No positive pairs found in batch! Batch size: {embeddings.shape[0]}, "
                f"Unique songs: {len(torch.unique(song_labels))}, "
                f"This likely means each song only appears once in the batch.")
        return s / c
