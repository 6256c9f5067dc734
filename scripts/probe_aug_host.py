import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("threads", torch.get_num_threads(), "cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
from mst_amd.mixing_utils import AudioAugmenter
from mst_amd.synth import synth_batch
a = AudioAugmenter()
torch.manual_seed(0)
a.draw_decisions(2)
for nt in (None, 16, 4, 1):
    if nt: torch.set_num_threads(nt)
    t = time.perf_counter(); d = a.draw_decisions(24); t1 = time.perf_counter() - t
    t = time.perf_counter(); [a._make_ir() for _ in range(24)]; t2 = time.perf_counter() - t
    t = time.perf_counter(); [torch.rand(1) < 0.5 for _ in range(400)]; t3 = time.perf_counter() - t
    print(f"threads={torch.get_num_threads()}: draw24 {t1*1e3:.1f} ms, 24 IR {t2*1e3:.1f} ms, 400 rand {t3*1e3:.1f} ms")
x = synth_batch(24, 441000, device="cuda")
st = {s: x[:, 2*i:2*i+2] for i, s in enumerate(("vocals","bass","drums","other"))}
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    y = a.augment_stems(st, decisions=d); torch.cuda.synchronize()
    print(f"augment_stems with ready decisions: {(time.perf_counter()-t)*1e3:.1f} ms")
