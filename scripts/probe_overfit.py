"""Overfit ONE fixed batch with the training step (dropout on): the loss must fall for either small-nets backend."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd.loss import InfoNCELoss
from mst_amd.mixing_utils import deferred_features
from mst_amd.model import MixingStyleEncoder
from mst_amd.synth import synth_batch

B, T = 12, 44100
for backend in ("hip", "torch"):
    for prec in ("fp32", "f16"):
        torch.manual_seed(0)
        model = MixingStyleEncoder(44100, 1024, 256, 128, 20, 10, 8, 768, feature_dim=64).cuda().train()
        model.train_backend, model.train_precision, model.small_nets_backend = "hip-strict", prec, backend
        opt = torch.optim.AdamW(model.parameters(), lr=float(os.environ.get("LR", "1e-3")))
        x = synth_batch(B, T, device="cuda")
        stems = {s: x[:, 2 * i:2 * i + 2] for i, s in enumerate(("vocals", "bass", "drums", "other"))}
        labels = torch.arange(B, device="cuda") // 2
        deferred = torch.stack([deferred_features(64)] * B).cuda()
        crit = InfoNCELoss(0.1)
        out = []
        for step in range(40):
            loss = crit(model(stems, deferred), labels)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            out.append(loss.item())
        print(backend, prec, " ".join(f"{v:.3f}" for v in out[::4]), flush=True)
