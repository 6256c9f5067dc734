import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd.mixing_utils import MelFeatPlan
from mst_amd.synth import synth_batch
T, B = 441000, 72
x = synth_batch(B, T, device="cuda")
plan = MelFeatPlan(44100, 1024, 256, 128)
for _ in range(3): plan.forward(x, True, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 30; e0.record()
for _ in range(n): plan.forward(x, True, True)
e1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('MST_LIB','default')}: {e0.elapsed_time(e1) / n:.3f} ms")
