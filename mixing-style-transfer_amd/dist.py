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
    def __init__(self, model, group=None, average=False):
        self.model, self.group, self.average = model, group, average
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.params = {b: [] for b in BUCKETS}
        for n, p in model.named_parameters():
            if p.requires_grad:
                self.params[bucket_of(n)].append(p)
        self._bucket_of = {id(p): b for b, ps in self.params.items() for p in ps}
        self._ready = {b: 0 for b in BUCKETS}
        self._pending = []           # (handle, flat buffer, parameters) of launched buckets
        self._early = set()          # buckets whose reduction the trunk launched itself on its stacked gradients
        self.launched = []           # bucket names in launch order (tests / logging)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for ps in self.params.values() for p in ps]
        model._grad_reducer = self   # the hand-written trunk looks here (model._HipTrunk.backward)

    def close(self):
        for h in self._hooks:
            h.remove()
        if getattr(self.model, "_grad_reducer", None) is self:
            self.model._grad_reducer = None

    # -- called from autograd ------------------------------------------------------------------------------------
    def _on_grad(self, p):
        b = self._bucket_of[id(p)]
        self._ready[b] += 1
        if self._ready[b] == len(self.params[b]) and b not in self._early:
            self._launch(b)

    def _launch(self, b):
        ps = [p for p in self.params[b] if p.grad is not None]
        if not ps or self.world == 1:
            self.launched.append(b)
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        h = dist.all_reduce(flat, group=self.group, async_op=True)
        self._pending.append((h, flat, ps))
        self.launched.append(b)

    def reduce_stacked(self, b, tensors):
        """The trunk's early launch: `tensors` are its STACKED gradient tensors of bucket b (the per-band .grad tensors will be
        views of them), reduced in place while the rest of the backward runs."""
        self._early.add(b)
        self.launched.append(b)
        if self.world == 1:
            return
        flat = torch.cat([t.reshape(-1) for t in tensors])
        h = dist.all_reduce(flat, group=self.group, async_op=True)
        self._pending.append((h, flat, list(tensors)))

    # -- called by the trainer before optimizer.step() -------------------------------------------------------------
    def wait(self):
        for h, flat, dst in self._pending:
            h.wait()
            if self.average:
                flat.div_(self.world)
            off = 0
            for t in dst:
                g = t.grad if isinstance(t, torch.nn.Parameter) else t
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g))
                off += n
        self._pending.clear()
        self._early.clear()
        for b in BUCKETS:
            self._ready[b] = 0
        order, self.launched = self.launched, []
        return order
