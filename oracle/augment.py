"""Oracle: AudioAugmenter (reference src/mixing_utils.py:364-479), CPU.

Same RNG consumption order on the global torch CPU generator as the reference:
per stem [coin_gain,(gain)] [coin_tilt,(hi/lo coin)] [coin_comp] [coin_bw,(cutoff)],
then [coin_reverb,(randn(L))].  Returns the augmented stems and a decision trace.
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import butter, sosfilt

STEMS = ("vocals", "bass", "drums", "other")


def _sosfilt32(sos, x: torch.Tensor) -> torch.Tensor:
    return torch.from_numpy(sosfilt(sos, x.cpu().numpy(), axis=-1)).float()


def compress(x: torch.Tensor, threshold=-20.0, ratio=4.0) -> torch.Tensor:
    """mixing_utils.py:435-447"""
    db = 20 * torch.log10(x.abs() + 1e-8)
    db2 = torch.where(db > threshold, threshold + (db - threshold) / ratio, db)
    return torch.sign(x) * (10 ** (db2 / 20))


def reverb(x: torch.Tensor, ir: torch.Tensor, wet=0.3) -> torch.Tensor:
    """mixing_utils.py:458-479: cross-correlation with ir, pad L//2, keep [:T]."""
    L = ir.numel()
    y = F.conv1d(x[:, None, :], ir[None, None, :], padding=L // 2)[:, 0, :]
    return x * (1 - wet) + y[:, :x.shape[1]] * wet


def make_ir(sr: int, decay=0.5) -> torch.Tensor:
    L = int(sr * decay)
    t = torch.linspace(0, decay, L)
    return torch.exp(-t / (decay / 4)) * torch.randn(L) * 0.1


def augment_stems(stems_dict, sr=44100, gain_range=9.0, prob=0.5):
    """stems_dict {stem: (2,T)} -> (aug dict, trace dict)."""
    out, trace = {}, {}
    for name, x in stems_dict.items():
        x = x.clone()
        t = {}
        if torch.rand(1) < prob:
            gdb = torch.rand(1) * 2 * gain_range - gain_range
            t["gain_db"] = float(gdb)
            x = x * (10 ** (gdb / 20))
        if torch.rand(1) < prob:
            hi = bool(torch.rand(1) < 0.5)
            t["tilt"] = "high" if hi else "low"
            sos = butter(2, 2000, btype="high", fs=sr, output="sos") if hi else \
                butter(2, 500, btype="low", fs=sr, output="sos")
            x = _sosfilt32(sos, x)
        if torch.rand(1) < prob:
            t["comp"] = True
            x = compress(x)
        if torch.rand(1) < prob:
            fc = torch.rand(1) * 8000 + 4000
            t["cutoff"] = fc.item()
            x = _sosfilt32(butter(4, fc.item(), btype="low", fs=sr, output="sos"), x)
        out[name], trace[name] = x, t
    if torch.rand(1) < prob:
        mix = sum(out.values())
        ir = make_ir(sr)
        trace["reverb_ir"] = ir
        mix = reverb(mix, ir)
        tot = sum([torch.mean(s ** 2) for s in out.values()]) + 1e-8
        for name in out:
            out[name] = out[name] + mix * (torch.mean(out[name] ** 2) / tot) * 0.3
    return out, trace
