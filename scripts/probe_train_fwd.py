import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd.mixing_utils import MixingFeatureExtractor
from mst_amd.model import MixingStyleEncoder
from mst_amd.synth import synth_batch
B, T = 72, 441000
torch.manual_seed(0)
model = MixingStyleEncoder(44100, 1024, 256, 128, 20, 10, 8, 768, feature_dim=64).cuda().eval()
fe = MixingFeatureExtractor()
x = synth_batch(B, T, device="cuda")
stems = {s: x[:, 2 * i:2 * i + 2] for i, s in enumerate(("vocals", "bass", "drums", "other"))}
with torch.no_grad():
    feats, lm = fe.features_and_logmel(stems)
    enc = model.hip_encoder()
    for name, fn in (("eval", lambda: enc.forward(lm, feats)), ("train-fwd", lambda: enc.forward_train(lm, feats))):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 5:.2f} ms")
