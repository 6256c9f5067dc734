"""The IIR chain kernel alone under fixed decision sets (no reverb): ms per call for 24 clips of 10 s."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd import _lib
from mst_amd.mixing_utils import AudioAugmenter
from mst_amd.synth import synth_batch
B = 24
x = synth_batch(B, 441000, device="cuda")
aug = AudioAugmenter(44100, 9.0, 0.5)
def decisions(gain, tilt, comp, bw, frac=1.0):
    clips = (_lib.AugClip * B)()
    for b in range(B):
        for s in range(4):
            st = clips[b].stem[s]
            st.gain = 1.0
            if (b * 4 + s) % 100 >= frac * 100:
                continue
            if gain: st.gain = 1.5
            if tilt: aug._draw_tilt(st, {})
            if comp: st.compress = 1
            if bw: aug._draw_bw(st, {})
    return clips, [None] * B, [{}] * B
y = x.clone()
out = []
for name, args in (("gain", (1, 0, 0, 0)), ("comp", (0, 0, 1, 0)), ("tilt", (0, 1, 0, 0)), ("bw", (0, 0, 0, 1)), ("all", (1, 1, 1, 1)), ("all/half", (1, 1, 1, 1, 0.5))):
    torch.manual_seed(3)
    dec = decisions(*args)
    for _ in range(2):
        aug.augment_packed_(y, decisions=dec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        aug.augment_packed_(y, decisions=dec)
    e1.record()
    torch.cuda.synchronize()
    out.append(f"{name} {e0.elapsed_time(e1) / 5:.3f}")
print(f"{os.path.basename(os.environ.get('MST_LIB', 'default'))}: " + "  ".join(out) + " ms", flush=True)
