"""Stage A alone at the contract size (72 clips of 10 s), every output layout the bench uses: ms per launch (HIP events).
MST_LIB selects an A/B build (variants/)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd import _lib
from mst_amd.mixing_utils import MelFeatPlan
from mst_amd.synth import synth_batch
T, B = 441000, 72
x = synth_batch(B, T, device="cuda")
plan = MelFeatPlan(44100, 1024, 256, 128)
stems = {s: x[:, 2 * i:2 * i + 2] for i, s in enumerate(("vocals", "bass", "drums", "other"))}
out = []
for name, lay, lo in (("cm32", _lib.LOGMEL_CM32, True), ("cm16", _lib.LOGMEL_CM16, True), ("cm16hi", _lib.LOGMEL_CM16, False), ("ref", _lib.LOGMEL_REF, True)):
    f = lambda: plan.forward_stems(stems, True, True, lay, want_absmax=lay == _lib.LOGMEL_CM16, want_lo=lo)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    out.append(f"{name} {e0.elapsed_time(e1) / n:.3f}")
print(f"{os.path.basename(os.environ.get('MST_LIB', 'default'))}: " + "  ".join(out) + " ms", flush=True)
