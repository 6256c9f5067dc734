"""Quick timing probe of stage A on the GPU box (not the contract bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd.mixing_utils import MelFeatPlan
from mst_amd.synth import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 72
T = int(sys.argv[2]) if len(sys.argv) > 2 else 441000
x = synth_batch(B, T, device="cuda")
plan = MelFeatPlan(44100, 1024, 256, 128)
for want_lm in (True, False):
    for _ in range(3):
        plan.forward(x, want_lm, True)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    ev0.record()
    for _ in range(n):
        plan.forward(x, want_lm, True)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / n
    by = B * (8 * T * 4 + (8 * 128 * (1 + T // 256) * 4 if want_lm else 0) + 256)
    print(f"B={B} T={T} logmel={want_lm}: {ms:.3f} ms/step  {B/ms*1e3:.0f} clips/s  {by/ms/1e6:.1f} GB/s algorithmic "
          f"({by/ms/1e6/8000*100:.1f}% of 8 TB/s)")
