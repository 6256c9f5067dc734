"""Augmentation chain alone: 24 clips of 10 s (the negatives of the contract batch), seeded decisions; ms per call (HIP events)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd.mixing_utils import AudioAugmenter
from mst_amd.synth import synth_batch
x = synth_batch(24, 441000, device="cuda")
aug = AudioAugmenter(44100, 9.0, 0.5)
torch.manual_seed(7)
dec = [aug.draw_decisions(24) for _ in range(4)]
y = x.clone()
for k in range(3):
    y.copy_(x); aug.augment_packed_(y, decisions=dec[k % 4])
torch.cuda.synchronize()
n = 12
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
for k in range(n):
    y.copy_(x)
    evs[k][0].record()
    aug.augment_packed_(y, decisions=dec[k % 4])
    evs[k][1].record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in evs)
print(f"{os.path.basename(os.environ.get('MST_LIB', 'default'))}: aug chain median {ts[n // 2]:.3f} ms  min {ts[0]:.3f}  checksum {float(y.double().abs().sum()):.6f}", flush=True)
