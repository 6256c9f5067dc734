"""Stages of the contract eval forward (72 clips) by event pairs of the encoder's own taps, for one conv precision
(PROBE_PREC = fp32 | f16x3-all | f16): median ms of conv1 and conv2 over 12 launches."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd import _lib
from mst_amd.mixing_utils import MixingFeatureExtractor
from mst_amd.model import MixingStyleEncoder
from mst_amd.synth import synth_batch
prec = os.environ.get("PROBE_PREC", "fp32")
torch.manual_seed(42)
m = MixingStyleEncoder(44100, 1024, 256, 128, 20, 10, 8, 768, feature_dim=64).cuda().eval()
m.conv1_precision = prec
x = synth_batch(72, 441000, device="cuda")
stems = {s: x[:, 2 * i:2 * i + 2] for i, s in enumerate(("vocals", "bass", "drums", "other"))}
enc = m.hip_encoder()
lay = enc.preferred_layout()
plan = m.audio_encoder.mel_preprocessor.plan(0)
with torch.no_grad():
    lm, _ = plan.forward_stems(stems, True, False, lay, want_absmax=lay == _lib.LOGMEL_CM16, want_lo=enc.mode != 3)
feats = torch.randn(72, 64, device="cuda")
def run(n):
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(n)]
    for e in evs:
        for q in e: q.record()
        enc.forward(lm, feats, events=e)
    torch.cuda.synchronize()
    return evs
run(3)
evs = run(12)
c1 = sorted(e[1].elapsed_time(e[2]) for e in evs); c2 = sorted(e[2].elapsed_time(e[3]) for e in evs)
at = sorted(e[3].elapsed_time(e[4]) for e in evs); pp = sorted(e[4].elapsed_time(e[5]) for e in evs); fm = sorted(e[0].elapsed_time(e[1]) for e in evs)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MST_"))
print(f"{prec:10s} {tag:40s} conv1 median {c1[6]:.4f} min {c1[0]:.4f}   conv2 median {c2[6]:.4f} min {c2[0]:.4f}   film {fm[6]:.4f} attn {at[6]:.4f} pool+proj {pp[6]:.4f}", flush=True)
