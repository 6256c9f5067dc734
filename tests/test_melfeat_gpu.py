"""GPU parity: HIP STFT/mel/feature kernels (through the C ABI) vs the CPU oracle and the goldens.

Tolerances (north_star: mel / features within 1e-4 rel fp32):
  log-mel   |d| <= 1e-4 * max(1, |ref|)            (natural-log domain)
  features  |d| <= 1e-4 * |ref| + 2e-4             (dB / ratio features; atol covers values near 0)
"""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import features as ofeat
from oracle import mel as omel

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def fe(**kw):
    from mst_amd.mixing_utils import MixingFeatureExtractor
    return MixingFeatureExtractor(**kw)


def run(x8, ext):
    d = omel.tensor_to_stems_dict(x8.cuda())
    f, lm = ext.features_and_logmel(d)
    torch.cuda.synchronize()
    return f.cpu(), lm.cpu()


def scaled_err(a, ref):
    return ((a.double() - ref.double()).abs() / ref.double().abs().clamp(min=1.0)).max().item()


def check_logmel(lm, ref, tol=1e-4, x=None, cfg=None):
    """|gpu - ref| <= tol * max(1, |ref|).  When the input x is given the bound is made principled for
    ill-conditioned inputs: bins that sit > ~130 dB below the frame's spectral peak (a low-passed bass stem, a large DC
    offset) carry the fp32 FFT's rounding noise inside log(mel + 1e-10) in ANY fp32 implementation -- the reference's
    own CPU result is then up to 4e-3 away from the float64 result (real music, tests/golden/song_a.npz).  So:
      * where the fp32 oracle is itself within 0.2*tol of the float64-evaluated oracle: strict |gpu - oracle32| <= tol;
      * everywhere: the GPU is no further from float64 than the fp32 oracle is (max: 1.5x + tol, mean: 2x + tol/100)."""
    e32 = scaled_err(lm, ref)
    if x is None:
        assert e32 <= tol, f"log-mel max scaled err {e32:.3e}"
        return
    ref64 = omel.logmel(x.double(), *(cfg or ()))
    den = ref64.abs().clamp(min=1.0)
    g64 = (lm.double() - ref64).abs() / den
    r64 = (ref.double() - ref64).abs() / den
    g32 = (lm.double() - ref.double()).abs() / ref.double().abs().clamp(min=1.0)
    well = r64 <= 0.2 * tol
    assert well.float().mean().item() > 0.5, "oracle ill-conditioned almost everywhere?"
    assert g32[well].max().item() <= tol, f"log-mel err vs fp32 oracle on well-conditioned bins {g32[well].max().item():.3e}"
    assert g64.max().item() <= 1.5 * r64.max().item() + tol, \
        f"log-mel max err vs float64 {g64.max().item():.3e} (fp32 oracle's own: {r64.max().item():.3e})"
    assert g64.mean().item() <= 2.0 * r64.mean().item() + tol / 100, \
        f"log-mel mean err vs float64 {g64.mean().item():.3e} (fp32 oracle's own: {r64.mean().item():.3e})"


def check_feats(f, ref, rtol=1e-4, atol=2e-4):
    np.testing.assert_allclose(f.numpy().astype(np.float64), np.asarray(ref, dtype=np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", cases.FEATURE_CASES)
def test_features_edge_cases_vs_oracle_and_golden(name):
    g = np.load(os.path.join(G, "features.npz"))
    x = cases.feature_case(name, 44100)[None]
    f, lm = run(x, fe())
    rf, rmel = ofeat.extract_all_features(x, return_mel=True)
    check_logmel(lm, torch.log(rmel + 1e-10), x=x)
    check_feats(f[0], rf[0])
    check_feats(f[0], g[f"{name}.features"])


def test_logmel_goldens():
    g = np.load(os.path.join(G, "logmel.npz"))
    for name, T in (("synth", 4096), ("synth1", 22050), ("one_sided", 8192)):
        x = cases.feature_case(name, T)[None]
        _, lm = run(x, fe())
        assert tuple(lm.shape) == (1, 8, 128, 1 + T // 256)
        check_logmel(lm, torch.from_numpy(g[f"{name}_{T}.logmel"]))


def test_full_size_batch_vs_oracle_and_golden():
    """T=441000 (BASELINE clip size), B=3: oracle comparison, goldens, and batch independence."""
    g = np.load(os.path.join(G, "features.npz"))
    gl = np.load(os.path.join(G, "logmel.npz"))
    x = torch.stack([cases.synth_clip(c, 441000) for c in (0, 1, 5)], 0)
    f, lm = run(x, fe())
    assert tuple(lm.shape) == (3, 8, 128, 1723)
    rf, rmel = ofeat.extract_all_features(x, return_mel=True)
    check_logmel(lm, torch.log(rmel + 1e-10))
    check_feats(f, rf)
    check_feats(f[0], g["synth10s_0.features"])
    check_feats(f[1], g["synth10s_1.features"])
    idx = torch.from_numpy(gl["synth10s_0.logmel_idx"])
    check_logmel(lm[0].flatten()[idx], torch.from_numpy(gl["synth10s_0.logmel_samples"]))
    # size-independent property: a clip's outputs do not depend on its batch neighbours
    f1, lm1 = run(x[2:3], fe())
    assert torch.equal(f1[0], f[2]) and torch.equal(lm1[0], lm[2])


def test_second_config_detailed_and_odd_length():
    g = np.load(os.path.join(G, "features.npz"))
    gl = np.load(os.path.join(G, "logmel.npz"))
    x = cases.feature_case("synth1", 66150)[None]
    f, lm = run(x, fe(sample_rate=44100, n_fft=2048, hop_length=512, n_mels=80))
    check_feats(f[0], g["cfg2.features"])
    check_logmel(lm, torch.from_numpy(gl["cfg2.logmel"]))
    x = cases.feature_case("synth", 44100)[None]
    f, _ = run(x, fe(use_detailed_spectral=True, n_spectral_bins=32))
    assert f.shape[1] == 180
    check_feats(f[0], g["detailed.features"])
    x = cases.feature_case("synth1", 30001)[None]   # odd T: scalar (unaligned) load path
    f, lm = run(x, fe())
    check_feats(f[0], g["odd.features"])
    check_logmel(lm, omel.logmel(x))


@pytest.mark.parametrize("n_fft,hop,n_mels", [(512, 128, 64), (1024, 256, 256), (2048, 441, 96)])
def test_other_fft_sizes_vs_oracle(n_fft, hop, n_mels):
    x = cases.feature_case("synth1", 40000)[None]
    f, lm = run(x, fe(n_fft=n_fft, hop_length=hop, n_mels=n_mels))
    rf, rmel = ofeat.extract_all_features(x, 44100, n_fft, hop, n_mels, return_mel=True)
    check_logmel(lm, torch.log(rmel + 1e-10))
    check_feats(f[0], rf[0])


def test_linearity_and_channel_permutation_properties():
    """Full-size properties that need no oracle: scaling the waveform by 2 adds log(4) to every log-mel bin that is
    not at the 1e-10 floor; swapping L/R of every stem swaps the channel pairs and negates ILD."""
    x = cases.synth_clip(7, 441000)[None]
    ext = fe()
    f1, lm1 = run(x, ext)
    f2, lm2 = run(2.0 * x, ext)
    big = lm1 > -10.0
    assert (lm2[big] - lm1[big] - np.log(4.0)).abs().max().item() < 2e-4
    xs = x.clone()
    xs[:, 0::2], xs[:, 1::2] = x[:, 1::2], x[:, 0::2]
    f3, lm3 = run(xs, ext)
    assert torch.equal(lm3[:, 0::2], lm1[:, 1::2]) and torch.equal(lm3[:, 1::2], lm1[:, 0::2])
    for blk in (0, 15, 34, 49):   # ILD sits at offset 12 of each stem block
        assert abs(f3[0, blk + 12].item() + f1[0, blk + 12].item()) < 1e-4


def test_sub_methods_and_errors():
    from mst_amd import _lib
    ext = fe()
    x = cases.feature_case("white", 44100)
    d = {k: v.cuda() for k, v in omel.tensor_to_stems_dict(x).items()}
    full = ext.extract_all_features(d).cpu()
    assert tuple(full.shape) == (64,)
    assert torch.allclose(ext.extract_masking(d).cpu(), full[30:34])
    dyn = ext.extract_dynamics(d["bass"]).cpu()
    assert torch.allclose(dyn[:4], full[0:4], rtol=1e-6, atol=1e-6)
    with pytest.raises(_lib.MstError):
        ext.extract_all_features({k: v[:, :300] for k, v in d.items()})   # T <= n_fft/2: reflect pad impossible
    with pytest.raises(_lib.MstError):
        fe(n_fft=4096, hop_length=1024).plan()                             # unsupported FFT size
