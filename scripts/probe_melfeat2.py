import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd.mixing_utils import MelFeatPlan
from mst_amd.synth import synth_batch
T = 441000
for B in (4, 16, 72):
    x = synth_batch(B, T, device="cuda")
    plan = MelFeatPlan(44100, 1024, 256, 128)
    for _ in range(3): plan.forward(x, True, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20; e0.record()
    for _ in range(n): plan.forward(x, True, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"batches={os.environ.get('MST_MELFEAT_BATCHES','2')} B={B}: {ms:.3f} ms  {ms/B*1e3:.1f} us/clip")
