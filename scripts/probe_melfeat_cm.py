"""13 stage-A launches (3 warm-up + 10) at the contract size with the channel-minor fp32 output the contract path uses; for PMC passes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd import _lib
from mst_amd.mixing_utils import MelFeatPlan
from mst_amd.synth import synth_batch
x = synth_batch(72, 441000, device="cuda")
plan = MelFeatPlan(44100, 1024, 256, 128)
stems = {s: x[:, 2 * i:2 * i + 2] for i, s in enumerate(("vocals", "bass", "drums", "other"))}
lay = {"cm32": _lib.LOGMEL_CM32, "cm16": _lib.LOGMEL_CM16, "ref": _lib.LOGMEL_REF}[os.environ.get("MST_PROBE_LAYOUT", "cm32")]
for _ in range(13):
    plan.forward_stems(stems, True, True, lay, want_absmax=lay == _lib.LOGMEL_CM16)
torch.cuda.synchronize()
