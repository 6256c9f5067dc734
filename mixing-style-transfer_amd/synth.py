"""Synthetic 4-stem stereo clips for benchmarks and parity tests.

Layout [B, 8, T] fp32, channel order vL,vR,bL,bR,dL,dR,oL,oR.  Clip c is drawn from
`torch.Generator(device).manual_seed(seed + c)`:
  vocals = 0.05*N(0,1), first 20 % of the clip exactly 0 (2 s of a 10 s clip)
  bass   = mono 0.2*sin(2*pi*55*t) + 0.01*N(0,1)   (L == R)
  drums  = 0.1*N(0,1)*exp(-((t mod 0.5 s)/0.08))
  other  = 0.08*N(0,1), independent L/R, R scaled by 0.7
all clamped to [-1, 1].  CPU and GPU generators give different streams: parity tests
generate on CPU and copy; bench.py generates on the device.
"""
import math

import torch


def synth_clip(c: int, T: int, sample_rate: int = 44100, device="cpu", seed: int = 42) -> torch.Tensor:
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed + c)
    n = torch.randn(8, T, generator=g, device=dev, dtype=torch.float32)
    t = torch.arange(T, device=dev, dtype=torch.float32) / sample_rate
    x = torch.empty(8, T, device=dev, dtype=torch.float32)
    x[0:2] = 0.05 * n[0:2]
    x[0:2, : int(0.2 * T)] = 0.0
    bass = 0.2 * torch.sin(2 * math.pi * 55.0 * t) + 0.01 * n[2]
    x[2] = bass
    x[3] = bass
    x[4:6] = 0.1 * n[4:6] * torch.exp(-(torch.remainder(t, 0.5) / 0.08))
    x[6] = 0.08 * n[6]
    x[7] = 0.7 * 0.08 * n[7]
    return x.clamp_(-1.0, 1.0)


def synth_batch(B: int, T: int, sample_rate: int = 44100, device="cpu", seed: int = 42,
                first_clip: int = 0) -> torch.Tensor:
    return torch.stack([synth_clip(first_clip + c, T, sample_rate, device, seed) for c in range(B)], 0)
