"""conv1 of the contract forward alone (72 clips, channel-minor log-mel): ms per launch by event pairs of the encoder's own taps."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd import _lib
from mst_amd.mixing_utils import MixingFeatureExtractor
from mst_amd.model import MixingStyleEncoder
from mst_amd.synth import synth_batch
torch.manual_seed(42)
m = MixingStyleEncoder(44100, 1024, 256, 128, 20, 10, 8, 768, feature_dim=64).cuda().eval()
x = synth_batch(72, 441000, device="cuda")
stems = {s: x[:, 2 * i:2 * i + 2] for i, s in enumerate(("vocals", "bass", "drums", "other"))}
fe = MixingFeatureExtractor()
feats, lm = fe.features_and_logmel(stems, _lib.LOGMEL_CM32, False)
enc = m.hip_encoder()
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
print(f"conv1 group={os.environ.get('MST_CONV1_CLIP_GROUP', 'default(4)') if not os.environ.get('MST_CONV1_BAND_MAJOR') else 'band-major'}: median {c1[6]:.4f} ms min {c1[0]:.4f}   conv2 median {c2[6]:.4f}", flush=True)
