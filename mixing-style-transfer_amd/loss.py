"""InfoNCE loss with the reference semantics (src/loss.py:10-136), vectorised (no per-anchor host syncs) and
sharded: with torch.distributed initialised, embeddings and labels cross RCCL/xGMI in ONE packed all-gather and every
rank evaluates the (tiny) loss on the whole gathered batch -- no other collective in forward or backward
(SURVEY.md section 8e).  On CUDA/HIP tensors forward and backward run in libmst.so (`mst_infonce_forward/backward`);
CPU tensors (the gloo tests of the sharding logic) take the same formulas in torch ops."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _AllGatherWithGrad(torch.autograd.Function):
    """all-gather of a (n, K) tensor; backward returns this rank's slice of the gradient of the gathered tensor.
    `reduce_grad=True` first all-reduces that gradient -- needed when every rank differentiates only ITS OWN rows of the
    loss, so that the local embeddings also receive the terms where they act as columns of other ranks' anchors.
    InfoNCELoss does not need it: every rank evaluates the whole (replicated) loss on the gathered batch."""

    @staticmethod
    def forward(ctx, x, reduce_grad):
        import torch.distributed as dist
        ws, ctx.rank, ctx.n, ctx.reduce_grad = dist.get_world_size(), dist.get_rank(), x.shape[0], reduce_grad
        out = torch.empty(ws * ctx.n, *x.shape[1:], dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous())
        return out

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        if ctx.reduce_grad:
            g = g.contiguous().clone()
            dist.all_reduce(g)
        return g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n], None


def gather_embeddings(emb: torch.Tensor, labels: torch.Tensor, reduce_grad=True):
    """ONE all-gather over RCCL/xGMI of (N_local, D) fp32 embeddings with the int64 labels riding along as two extra
    fp32 columns (bit patterns; the collective only copies bytes) -> global embeddings and labels in rank order, plus
    the row offset of the local slice.  Differentiable w.r.t. `emb` (see _AllGatherWithGrad for `reduce_grad`)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return emb, labels, 0
    n, D = emb.shape
    bits = labels.to(torch.int64).contiguous().view(torch.float32).view(n, 2)
    packed = torch.cat([emb.float(), bits.to(emb.device)], dim=1)
    if emb.requires_grad:
        allp = _AllGatherWithGrad.apply(packed, reduce_grad)
    else:
        allp = torch.empty(dist.get_world_size() * n, D + 2, dtype=torch.float32, device=emb.device)
        dist.all_gather_into_tensor(allp, packed)
    all_l = allp[:, D:].detach().contiguous().view(torch.int64).view(-1).to(labels.dtype)
    return allp[:, :D], all_l, dist.get_rank() * n


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
    """Drop-in for reference InfoNCELoss(temperature)(embeddings (N, D), song_labels (N,)) -> scalar.

    `check`: how the reference's "No positive pairs found in batch!" guard (src/loss.py) is evaluated on the GPU path.
      "sync" (default)  -- as the reference: the count of anchors with a positive is read back and the RuntimeError raised before
                           the call returns.  One device -> host read per call: the host waits for the whole step and the GPU
                           idles for the launch latency of the next step's first kernels (~0.2 ms per step on this chip).
      "deferred"        -- the count goes to a pinned host word with an asynchronous copy; it is examined at the NEXT call (and
                           by `finish()`), by which time the next step's kernels are queued: the same RuntimeError, one call
                           later, no idle GPU.  The step without positives itself returns a zero loss with zero gradients
                           (sum 0 / max(count, 1)).  That step is NOT a no-op for the trainer's state: `optimizer.step()` on
                           zero gradients still applies weight decay and moves Adam's moments, and a training forward has
                           already updated the BatchNorm running statistics -- the model is one decay / momentum step past
                           the last good batch when the error surfaces.  `finish()` is therefore MANDATORY after the last
                           step of a loop and before every checkpoint (otherwise the last call's error is never raised);
                           the owed error of the previous call is raised at the top of the next call, before any of that
                           call's kernels are queued."""

    def __init__(self, temperature=0.1, gather=False, check="sync"):
        super().__init__()
        if check not in ("sync", "deferred"):
            raise ValueError("check must be 'sync' or 'deferred'")
        self.temperature = temperature
        self.gather = gather
        self.check = check
        self._pending = None   # (pinned count, event, message) of the previous call
        self.gather_events = None   # optional (start, end) torch.cuda.Event pair recorded around the all-gather (bench.py, N > 1)

    def finish(self):
        """Deferred mode: examine the last call's guard now (waits for that call's kernels); raises the RuntimeError it owes."""
        pend, self._pending = self._pending, None
        if pend is not None:
            flag, ev, msg = pend
            ev.synchronize()
            if flag.item() == 0:
                raise RuntimeError(msg)

    def forward(self, embeddings, song_labels):
        if self.check == "deferred" and embeddings.is_cuda:
            self.finish()   # the PREVIOUS call's guard, before this call queues anything (its kernels are at most one step behind)
        if self.gather:
            # sharded batch: one packed all-gather, then EVERY rank evaluates the whole loss on the gathered batch
            # (N^2 D flops: negligible next to the encoder).  No all-reduce in the forward, none in the backward:
            # the local slice of d loss / d all_embeddings is already complete, and the value is identical on all
            # ranks.  Parameter gradients must be SUMMED over ranks (each rank holds the part that flows through
            # its own clips).
            if self.gather_events is not None:
                self.gather_events[0].record()
            all_e, all_l, _ = gather_embeddings(embeddings, song_labels, reduce_grad=False)
            if self.gather_events is not None:
                self.gather_events[1].record()
            embeddings, song_labels = all_e, all_l
        rows_fn = info_nce_rows_hip if embeddings.is_cuda else info_nce_rows
        s, c = rows_fn(embeddings, song_labels, 0, embeddings.shape[0], self.temperature)
        if self.check == "deferred" and embeddings.is_cuda:
            bufs = self.__dict__.setdefault("_flags", [torch.zeros(1).pin_memory(), torch.zeros(1).pin_memory()])
            self._turn = 1 - getattr(self, "_turn", 0)
            flag = bufs[self._turn]
            flag.copy_(c.detach().reshape(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (flag, ev, f"No positive pairs found in batch! Batch size: {embeddings.shape[0]}, "
                                       "This likely means each song only appears once in the batch.")
            return s / torch.clamp(c, min=1.0)
        if c.item() == 0:
            raise RuntimeError(
                f"No positive pairs found in batch! Batch size: {embeddings.shape[0]}, "
                f"Unique songs: {len(torch.unique(song_labels))}, "
                f"This likely means each song only appears once in the batch.")
        return s / c
