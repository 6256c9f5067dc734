"""Are the gradient differences between the HIP training trunk and PyTorch fp32 autograd real, or the conditioning of
the problem?  Compares both against PyTorch float64 autograd of the same model (reference's train_baseline.sh shapes)."""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import cases
from oracle import mel as omel
from test_encoder_gpu import build_model

cfg = cases.CFG_BASELINE_SH
model, sd = build_model(cfg)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
ref32 = copy.deepcopy(model)
ref64 = copy.deepcopy(model).double()
for m_ in (model, ref32, ref64):
    m_.train()
model.train_backend, ref32.train_backend, ref64.train_backend = "hip", "torch", "torch"
B, T = 4, 44100
x = torch.stack([cases.synth_clip(c, T) for c in range(B)], 0).cuda()
stems = omel.tensor_to_stems_dict(x)
g = torch.Generator().manual_seed(8)
feats = (torch.randn(B, 64, generator=g) * 2.0).cuda()
R = torch.randn(B, cfg["embed_dim"], generator=g).cuda()
with torch.no_grad():
    lm = model.audio_encoder.mel_preprocessor(stems)
(model.forward_from_logmel(lm, feats) * R).sum().backward()
(ref32.forward_from_logmel(lm, feats) * R).sum().backward()
(ref64.forward_from_logmel(lm.double(), feats.double()) * R.double()).sum().backward()
rows = []
for (n, pa), (_, pb), (_, pc) in zip(model.named_parameters(), ref32.named_parameters(), ref64.named_parameters()):
    den = pc.grad.abs().max().item()
    if den < 1e-12 or n.endswith(("conv1.bias", "conv2.bias", "attention.2.bias")):
        continue
    rows.append((n, (pa.grad.double() - pc.grad).abs().max().item() / den, (pb.grad.double() - pc.grad).abs().max().item() / den))
rows.sort(key=lambda r: -max(r[1], r[2]))
print("parameter: |hip - f64| / max, |torch32 - f64| / max")
for r in rows[:10]:
    print(f"{r[0]:55s} {r[1]:.2e} {r[2]:.2e}")
