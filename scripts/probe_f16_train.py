"""Scratch: f16 training gradients -- HIP trunk vs the float64 oracle, and the fp32-evaluated oracle vs the float64 one."""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cases
from oracle import mel as omel
from oracle.train_f16 import convert_convs
from test_encoder_gpu import build_model

T = int(sys.argv[1]) if len(sys.argv) > 1 else 44100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = cases.CFG_DEFAULT
model, sd = build_model(cfg)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
ref32 = convert_convs(copy.deepcopy(model))
ref64 = convert_convs(copy.deepcopy(model).double())
for m_ in (model, ref32, ref64):
    m_.train()
model.train_backend, ref32.train_backend, ref64.train_backend = "hip-strict", "torch", "torch"
model.train_precision = "f16"
x = torch.stack([cases.synth_clip(c % 4, T) * (1.0 + 0.1 * (c // 4)) for c in range(B)], 0).cuda()
stems = omel.tensor_to_stems_dict(x)
g = torch.Generator().manual_seed(8)
feats = (torch.randn(B, 64, generator=g) * 2.0).cuda()
R = torch.randn(B, cfg["embed_dim"], generator=g).cuda()
with torch.no_grad():
    lm = model.audio_encoder.mel_preprocessor(stems)
la = (model.forward_from_logmel(lm, feats) * R).sum()
lb = (ref32.forward_from_logmel(lm, feats) * R).sum()
lc = (ref64.forward_from_logmel(lm.double(), feats.double()) * R.double()).sum()
la.backward(), lb.backward(), lc.backward()
print("loss", la.item(), lb.item(), lc.item())
for (n, pa), (_, pb), (_, pc) in zip(model.named_parameters(), ref32.named_parameters(), ref64.named_parameters()):
    den = pc.grad.abs().max().item()
    if den == 0 or n.endswith(("conv1.bias", "conv2.bias")):
        continue
    a = (pa.grad.double() - pc.grad).abs().max().item() / den
    b = (pb.grad.double() - pc.grad).abs().max().item() / den
    if a > 1e-4 or b > 1e-4:
        print(f"{n:55s} hip {a:.1e}  oracle32 {b:.1e}")
