"""Where does the stage-A log-mel error sit relative to the unavoidable fp32 FFT rounding floor?
z = |logmel - logmel_f64| * (mel64 + 1e-10) / (2 nu sqrt(sfb * mel64) + nu^2 sfb),  nu = eps32 * ||x_frame * w||_2
for the GPU kernels and for the fp32 CPU oracle (torch.stft = pocketfft).  Run on the GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import cases
from oracle import mel as omel
from mst_amd.mixing_utils import MixingFeatureExtractor

x = cases.song_a_clips()[:1]
ext = MixingFeatureExtractor()
f, lm = ext.features_and_logmel(omel.tensor_to_stems_dict(x.cuda()))
lm = lm.cpu().double()
o32 = omel.logmel(x).double()
mel64 = omel.mel_power(x.double())
r64 = torch.log(mel64 + 1e-10)
n_fft, hop = 1024, 256
w = omel.hann_periodic(n_fft).double()
xp = torch.nn.functional.pad(x.double(), (n_fft // 2, n_fft // 2), mode="reflect")
fr = xp.unfold(-1, n_fft, hop) * w                        # (B, 8, F, n_fft)
nu = (fr.pow(2).sum(-1).sqrt() * 2.0 ** -24)[:, :, None, :]  # (B, 8, 1, F)
fb = omel.htk_fbank(44100, n_fft, 128).double()           # (513, 128)
sfb = fb.sum(0)[None, None, :, None]
unit = (2 * nu * (sfb * mel64).sqrt() + nu * nu * sfb) / (mel64 + 1e-10)
for name, a in (("gpu", lm), ("oracle32", o32)):
    d = (a - r64).abs()
    z = d / unit.clamp(min=1e-30)
    sel = d > 2e-5
    print(name, "max|d|", d.max().item(), "z max", z[sel].max().item() if sel.any() else 0.0,
          "z rms", z.pow(2).mean().sqrt().item(), "z 99.9%", z.flatten().kthvalue(int(0.999 * z.numel())).values.item(),
          "frac d>1e-4", (d > 1e-4).double().mean().item())
d = (lm - o32).abs()
i = d.argmax(); idx = torch.unravel_index(i, d.shape)
print("worst gpu-vs-o32", d.max().item(), [int(k) for k in idx], "r64", r64[idx].item(), "gpu", lm[idx].item(), "o32", o32[idx].item(),
      "unit", unit[idx].item(), "framepeak log", r64[idx[0], idx[1], :, idx[3]].max().item())
d = (lm - r64).abs(); z = torch.where(d > 2e-5, d / unit.clamp(min=1e-30), torch.zeros_like(d))
for i in z.flatten().topk(8).indices:
    idx = torch.unravel_index(i, d.shape)
    print("top z", z[idx].item(), [int(k) for k in idx], "d", d[idx].item(), "mel64", mel64[idx].item(), "nu", nu[idx[0], idx[1], 0, idx[3]].item(),
          "gpu", lm[idx].item(), "o32", o32[idx].item(), "r64", r64[idx].item())
# per mel-band and per channel z statistics (bins above the 1e-10 floor)
ok = (mel64 > 1e-9) & (d > 0)
for c in range(8):
    zz = (d / unit.clamp(min=1e-30))[0, c][ok[0, c]]
    z3 = ((o32 - r64).abs() / unit.clamp(min=1e-30))[0, c][ok[0, c]]
    print("ch", c, "n", zz.numel(), "gpu z rms %.3f max %.1f | o32 z rms %.3f max %.1f" % (zz.pow(2).mean().sqrt().item(), zz.max().item(), z3.pow(2).mean().sqrt().item(), z3.max().item()))
