"""Debug: where does the HIP training trunk leave the float64 oracle on the training-fixture input? (GPU box)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from oracle import encoder as oenc, mel as omel
from test_encoder_gpu import build_model

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
cfg = cases.CFG_DEFAULT
B, T = 4, 66150
x = cases.pcm_batch(B, T)
g = np.load(os.path.join(ROOT, "tests/golden/train.npz"))
feats = torch.from_numpy(g[f"{tag}.features"])
R = torch.randn(B, cfg["embed_dim"], generator=torch.Generator().manual_seed(77))
# float64 oracle with taps
sd = {k: (v.double().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in cases.make_state_dict(cfg, seed=42).items()}
lm64 = omel.logmel(x.double())
taps = {}
emb = oenc.encoder_from_logmel(sd, lm64, feats.double(), 20, 10, taps=taps, bn_training=True)
(emb * R.double()).sum().backward()
# HIP
model, _ = build_model(cfg)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
model.train(); model.train_backend = "hip-strict"; model.train_precision = prec
stems = omel.tensor_to_stems_dict(x.cuda())
with torch.no_grad():
    lm = model.audio_encoder.mel_preprocessor(stems)
print("logmel err", (lm.cpu().double() - lm64).abs().max().item())
from mst_amd.model import HipEncoder
enc = HipEncoder(model, "fp32"); enc.set_train_precision(prec)
from test_encoder_gpu import _stacked_trunk_params
enc.update_trunk_params(*_stacked_trunk_params(model))
film = taps["film"].detach().float().cuda()
_, t = enc.forward_train(lm, film=film, head=False)
ns = 11
for i in range(ns):
    p1 = t["pool1"][:, i].cpu().double(); r1 = taps[f"pool1_{i}"].detach()
    p2 = t["pool_in"].cpu().double().reshape(B, ns, 64, 2, -1)[:, i]; r2 = taps[f"pool2_{i}"].detach()
    e1 = (p1 - r1).abs(); e2 = (p2 - r2).abs()
    # windows whose ReLU state differs
    flip1 = ((p1 > 0) != (r1 > 0)).sum().item(); flip2 = ((p2 > 0) != (r2 > 0)).sum().item()
    print(f"band {i}: pool1 err {e1.max().item():.2e} (max {r1.abs().max().item():.2f}) relu flips {flip1}; pool2 err {e2.max().item():.2e} flips {flip2}")
# gradient check through the model path
emb_h = model(stems, feats.cuda())
(emb_h * R.cuda()).sum().backward()
for n, q in model.named_parameters():
    if ".3." in n or ".9." in n:
        ref = sd[n].grad
        print(n, f"{(q.grad.cpu().double() - ref).abs().max().item() / ref.abs().max().item():.2e}")
