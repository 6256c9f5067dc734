"""Train-step timing (stage A + loss in HIP; encoder: TRAIN_BACKEND=hip (default: native trunk forward and
pool/ReLU/FiLM/BN backward, conv gradients via MIOpen) or TRAIN_BACKEND=torch (all PyTorch-ROCm)).  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd.loss import InfoNCELoss
from mst_amd.mixing_utils import MixingFeatureExtractor
from mst_amd.model import MixingStyleEncoder
from mst_amd.synth import synth_batch

B, T = int(os.environ.get("B", 72)), 441000
torch.manual_seed(0)
model = MixingStyleEncoder(44100, 1024, 256, 128, 20, 10, 8, 768, feature_dim=64).cuda().train()
model.train_backend = os.environ.get("TRAIN_BACKEND", "hip")   # "torch": encoder fwd/bwd on PyTorch-ROCm/MIOpen
model.train_precision = os.environ.get("TRAIN_PRECISION", "fp32")   # "f16" / "f16x3": see MixingStyleEncoder.train_precision
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
fe = MixingFeatureExtractor()
crit = InfoNCELoss(0.1)
x = synth_batch(B, T, device="cuda")
stems = {s: x[:, 2 * i:2 * i + 2] for i, s in enumerate(("vocals", "bass", "drums", "other"))}
labels = torch.arange(B, device="cuda") // 3

from mst_amd.mixing_utils import deferred_features
deferred = torch.stack([deferred_features(64)] * B).cuda()


def step():
    emb = model(stems, deferred)   # the trainer's call: one stage-A launch inside yields features + log-mel
    loss = crit(emb, labels)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss

for _ in range(int(os.environ.get("WARMUP", 2))):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"B={B}: train step {dt * 1e3:.1f} ms = {B / 3 / dt:.1f} triplets/s, loss {l.item():.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
from mst_amd import model as _m
if _m._HipTrunk.last_timing:
    print("backward sections (ms):", _m._HipTrunk.last_timing)

