"""Data-parallel gradient exchange for the contrastive training step (nothing of the kind exists in the
reference, which trains on one GPU: src/train.py:439,545-555).

One process per GPU; the clips shard over the ranks and every rank back-propagates the replicated InfoNCE through ITS
clips (loss.py), so the parameter gradients must be SUMMED over the ranks before the optimizer step.  `GradientReducer` does
that with a few bucketed all-reduces over RCCL that run WHILE the backward pass is still computing:

  head    audio_encoder.attention_pooling.*   ready first (its gradients come out of autograd before the trunk's backward)
  conv2   conv2 / bn2 of all sub-bands        launched by the hand-written trunk right after its conv2 weight gradient,
                                              i.e. behind ~60 % of the trunk's backward (conv2 input gradient, layer-1 backward
                                              and conv1 weight gradient still to come)
  conv1   conv1 / bn1 of all sub-bands        at the end of the trunk's backward
  film    film_encoder.*                      last (the FiLM MLP sits behind the trunk's FiLM gradient)

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a 13 MB gradient is 1.7 MB per link per ring phase -- the exchange is
latency-bound, so FEW collectives (one per bucket, flattened) matter more than their size.  Bucket reductions are launched with
async_op=True and waited for in `wait()`, which the trainer calls right before `optimizer.step()`.

    reducer = GradientReducer(model)          # once
    loss.backward(); reducer.wait(); optimizer.step(); optimizer.zero_grad(set_to_none=True)
"""
import torch
import torch.distributed as dist

BUCKETS = ("head", "conv2", "conv1", "film")


def bucket_of(name: str) -> str:
    if name.startswith("film_encoder."):
        return "film"
    if ".subnet_cnns." in name:
        return "conv2" if (".conv2." in name or ".bn2." in name) else "conv1"
    return "head"


class GradientReducer:
    """What is exchanged is always THIS backward's gradient (never `.grad`, which may already hold earlier contributions), and what
    `wait()` leaves in `.grad` is `old + reduced`, computed from values that are identical on every rank (`old` = what `.grad` held
    when the backward started, snapshotted by a tensor hook only when it existed; `reduced` = the all-reduced gradient of this
    backward) -- so the ranks' gradients stay BIT-IDENTICAL whatever `.grad` held: None (the fast path: autograd adopts the trunk's
    views, `reduced` is copied over them, no snapshot), zeros (`zero_grad(set_to_none=False)`), or the reduced gradients of
    earlier micro-batches (accumulation: backward, wait, backward, wait, step).  A second backward WITHOUT a `wait()` in
    between raises; a bucket in which some parameter received no gradient is reduced late, by `wait()` (`.late`)."""

    def __init__(self, model, group=None, average=False):
        self.model, self.group, self.average = model, group, average
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.params = {b: [] for b in BUCKETS}
        for n, p in model.named_parameters():
            if p.requires_grad:
                self.params[bucket_of(n)].append(p)
        self._bucket_of = {id(p): b for b, ps in self.params.items() for p in ps}
        self._ready = {b: 0 for b in BUCKETS}
        self._incoming = {}          # id(parameter) -> the gradient this backward is about to accumulate into .grad
        self._old = {}               # id(parameter) -> copy of a .grad that existed when this backward reached the parameter
        self._pending = []           # (handle, flat buffer, destinations, owners) of launched buckets
        self._early = set()          # buckets whose reduction the trunk launched itself on its stacked gradients
        self.launched = []           # bucket names in launch order (tests / logging)
        self.late = []               # buckets `wait()` had to launch itself last time (a parameter got no gradient)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for ps in self.params.values() for p in ps]
        self._hooks += [p.register_hook(lambda g, p=p: self._on_incoming(p, g)) for ps in self.params.values() for p in ps]
        model._grad_reducer = self   # the hand-written trunk looks here (model._HipTrunk.backward)

    def close(self):
        for h in self._hooks:
            h.remove()
        if getattr(self.model, "_grad_reducer", None) is self:
            self.model._grad_reducer = None

    # -- called from autograd ------------------------------------------------------------------------------------
    def _on_incoming(self, p, g):    # Tensor.register_hook: runs before AccumulateGrad adds g to p.grad
        self._incoming[id(p)] = g    # (returns None: the gradient itself is left as it is)
        if self.world > 1 and p.grad is not None and id(p) not in self._old:
            self._old[id(p)] = p.grad.detach().clone()

    def _on_grad(self, p):
        b = self._bucket_of[id(p)]
        self._ready[b] += 1
        if self._ready[b] > len(self.params[b]):
            raise RuntimeError(f"GradientReducer: a second backward reached bucket '{b}' before wait() collected the first one's "
                               "reductions; call reducer.wait() after every backward (gradient accumulation: backward, wait, backward, "
                               "wait, step)")
        if self._ready[b] == len(self.params[b]) and b not in self._early:
            self._launch(b)

    def _launch(self, b):
        ps = [p for p in self.params[b] if id(p) in self._incoming]
        self.launched.append(b)
        if not ps or self.world == 1:
            return
        inc = [self._incoming[id(p)] for p in ps]
        flat = torch.cat([g.reshape(-1) for g in inc])
        h = dist.all_reduce(flat, group=self.group, async_op=True)
        self._pending.append((h, flat, ps, inc))

    def reduce_stacked(self, b, tensors, owners=None):
        """The trunk's early launch: `tensors` are its STACKED gradient tensors of bucket b (this backward's gradients; the per-band
        gradients autograd receives are views of them), reduced while the rest of the backward runs.  owners[k][i] = the Parameter
        whose gradient is tensors[k][i]: where its `.grad` turns out NOT to be that view (autograd accumulated into an existing
        `.grad`), `wait()` adds `reduced - local` to it."""
        if b in self._early:
            raise RuntimeError(f"GradientReducer: a second backward reached bucket '{b}' before wait() collected the first one's reductions")
        self._early.add(b)
        self.launched.append(b)
        if self.world == 1:
            return
        flat = torch.cat([t.reshape(-1) for t in tensors])
        h = dist.all_reduce(flat, group=self.group, async_op=True)
        self._pending.append((h, flat, list(tensors), owners))

    def _write(self, params, reduced):
        """.grad <- old + reduced (old: the snapshot of a pre-existing .grad) or <- reduced: functions of rank-identical values only."""
        new = [p for p in params if id(p) not in self._old]
        if new:
            torch._foreach_copy_([p.grad for p in new], [r for p, r in zip(params, reduced) if id(p) not in self._old])
        acc = [(p, r) for p, r in zip(params, reduced) if id(p) in self._old]
        if acc:
            torch._foreach_copy_([p.grad for p, _ in acc], torch._foreach_add([self._old[id(p)] for p, _ in acc], [r for _, r in acc]))

    # -- called by the trainer before optimizer.step() -------------------------------------------------------------
    def wait(self):
        self.late = [b for b in BUCKETS if b not in self.launched and self.params[b]]
        for b in self.late:          # a parameter of the bucket got no gradient in this backward: reduce what did arrive, now
            self._launch(b)
        if self.world > 1 and set(self.launched) != {b for b in BUCKETS if self.params[b]}:
            raise RuntimeError(f"GradientReducer.wait(): buckets launched {self.launched}, expected {BUCKETS}")
        for h, flat, dst, extra in self._pending:
            h.wait()
            if self.average:
                flat.div_(self.world)
            off, views = 0, []
            for t in dst:
                n = t.numel()
                views.append(flat[off:off + n].view(t.shape))
                off += n
            if isinstance(dst[0], torch.nn.Parameter):   # hook bucket (extra: this backward's local gradients, unused here)
                self._write(dst, views)
                continue
            for k, (t, r) in enumerate(zip(dst, views)):   # trunk bucket: t = stacked local gradients, extra = owners
                t.copy_(r)                                 # every .grad that is a view of t (autograd adopted it) is done
                if extra is not None:
                    lo, hi = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
                    stray = [(p, i) for i, p in enumerate(extra[k]) if p.grad is not None and not lo <= p.grad.data_ptr() < hi]
                    if stray:   # autograd accumulated into an existing .grad instead of adopting the view
                        self._write([p for p, _ in stray], [r[i] for _, i in stray])
        self._pending.clear()
        self._early.clear()
        self._incoming.clear()
        self._old.clear()
        for b in BUCKETS:
            self._ready[b] = 0
        order, self.launched = self.launched, []
        return order
