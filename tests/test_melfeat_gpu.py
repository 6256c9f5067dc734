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
import parity
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


def fft_noise_unit(x, sample_rate=44100, n_fft=1024, hop=256, n_mels=128):
    """Per log-mel bin: the change of log(mel + 1e-10) caused by ONE unit of fp32 FFT rounding noise, from the input
    alone (float64).  A length-N fp32 FFT returns X[k] + e[k] with |e| ~ nu = 2^-24 * ||frame * window||_2 (times a
    constant of a few units that depends on the factorisation; up to ~100 in the tail, where a weak bin shares the
    last butterflies with the frame's strongest bin), so  d mel <= sum_k fb[k] (2 |X[k]| nu + nu^2)
    <= 2 nu sqrt(sum fb * mel) + nu^2 sum fb."""
    xd = x.double()
    mel64 = omel.mel_power(xd, sample_rate, n_fft, hop, n_mels)
    w = omel.hann_periodic(n_fft).double()
    xp = torch.nn.functional.pad(xd, (n_fft // 2, n_fft // 2), mode="reflect")
    nu = torch.empty(mel64.shape[:-2] + (1, mel64.shape[-1]), dtype=torch.float64)
    for b in range(xd.shape[0]):   # bounded memory: one clip at a time
        fr = xp[b].unfold(-1, n_fft, hop) * w
        nu[b, :, 0, :] = fr.pow(2).sum(-1).sqrt() * 2.0 ** -24
    sfb = omel.htk_fbank(sample_rate, n_fft, n_mels).double().sum(0)[None, None, :, None]
    unit = (2 * nu * (sfb * mel64).sqrt() + nu * nu * sfb) / (mel64 + 1e-10)
    return unit, torch.log(mel64 + 1e-10)


def check_logmel(lm, ref, tol=1e-4, x=None, cfg=None, name="log-mel"):
    """|gpu - ref| <= tol * max(1, |ref|).  When the input x is given the bound is made principled for
    ill-conditioned inputs: bins far below the frame's spectral peak (a low-passed bass stem of real music, a large DC
    offset) carry the fp32 FFT's rounding noise inside log(mel + 1e-10) in ANY fp32 implementation -- on real music
    (tests/golden/song_a_crops.npz) the reference's own CPU result is up to 0.12 (natural-log units) away from the
    float64 result and 8 % of its bins are off by more than 1e-4.  With unit = fft_noise_unit(x) (conditioning of
    each bin, computed from the input alone) and z = (|result - float64 result| - tol * max(1, |ref|)) / unit, the
    noise a result carries MEASURED in units:
      * where 64 * unit <= 0.2 * tol (well-conditioned bins): strict |gpu - fp32 oracle| <= tol * max(1, |ref|);
      * everywhere: max z_gpu <= 2 * max(max z_oracle32, 8) -- the kernel's worst bin needs at most twice the noise
        allowance the reference's own fp32 FFT (pocketfft) needs on the same input (an extreme-value statistic over
        ~10^6 bins of two different FFT factorisations: measured 0.6-1.6x; round 1 allowed a fixed 256 units);
      * over the bins with unit > 1e-5: rms(|gpu - f64| / unit) <= 2 * rms(|oracle32 - f64| / unit) + 1.
    Every call records the error distribution (tests/parity.py)."""
    import inspect
    fr = inspect.stack()[1]
    name = f"{fr.function.replace('test_', '')[:34]}:{fr.lineno} {name}"
    e32 = scaled_err(lm, ref)
    parity.record(name + " vs fp32 ref", lm, ref, floor_abs=1.0)
    if x is None:
        assert e32 <= tol, f"log-mel max scaled err {e32:.3e}"
        return
    unit, ref64 = fft_noise_unit(x, *(cfg or ()))
    den = ref64.abs().clamp(min=1.0)
    g64 = (lm.double() - ref64).abs()
    r64 = (ref.double() - ref64).abs()
    g32 = (lm.double() - ref.double()).abs()
    well = 64 * unit <= 0.2 * tol
    if well.any():
        assert (g32[well] / den[well]).max().item() <= tol, \
            f"log-mel err vs fp32 oracle on well-conditioned bins {(g32[well] / den[well]).max().item():.3e}"
    zg_max = ((g64 - tol * den) / unit).max().item()
    zr_max = ((r64 - tol * den) / unit).max().item()
    live = unit > 1e-5   # bins where FFT noise, not the rounding of log() itself, is what is being measured
    zg = zr = 0.0
    if live.any():
        zg = (g64[live] / unit[live]).pow(2).mean().sqrt().item()
        zr = (r64[live] / unit[live]).pow(2).mean().sqrt().item()
    # the literal 1e-4 bar as plain counts (GPU test log): which share of the bins is further than tol * max(1, |ref|) from the
    # FLOAT64 log-mel for the kernel and for the reference's own fp32 arithmetic (torch.stft / pocketfft), and from each other
    frac_g64 = float((g64 > tol * den).double().mean())
    frac_r64 = float((r64 > tol * den).double().mean())
    frac_g32 = float((g32 > tol * den).double().mean())
    parity.note(name + " FFT noise (units of 2^-24 |frame|)", gpu_max=zg_max, oracle32_max=zr_max, gpu_rms=zg,
                oracle32_rms=zr, well_conditioned_frac=float(well.double().mean()))
    parity.note(name + " share of bins beyond 1e-4", gpu_vs_float64=frac_g64, fp32_oracle_vs_float64=frac_r64,
                gpu_vs_fp32_oracle=frac_g32, bins=int(den.numel()))
    assert frac_g64 <= 1.25 * frac_r64 + 1e-3, \
        f"share of log-mel bins beyond {tol:g} of the float64 result: gpu {frac_g64:.4f}, the reference's fp32 arithmetic {frac_r64:.4f}"
    assert zg_max <= 2.0 * max(zr_max, 8.0), \
        f"log-mel worst-bin noise: gpu needs {zg_max:.1f} units, the fp32 oracle {zr_max:.1f}"
    assert zg <= 2.0 * zr + 1.0, f"rms noise (units): gpu {zg:.2f} vs fp32 oracle {zr:.2f}"


def check_feats(f, ref, rtol=1e-4, atol=2e-4, name="features"):
    import inspect
    fr = inspect.stack()[1]
    parity.record(f"{fr.function.replace('test_', '')[:34]}:{fr.lineno} {name}", f, ref, floor_abs=2.0)     # |d| <= 1e-4 |ref| + 2e-4  ==  rel <= 1e-4 with the floor at 2
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


@pytest.mark.parametrize("n_fft,hop,n_mels", [(512, 128, 64), (1024, 256, 256), (2048, 441, 96),
                                              (1024, 256, 80), (1024, 256, 40), (2048, 512, 128), (2048, 512, 48)])
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
    # L and R share one complex FFT (real / imaginary part), so the swap is exact only up to rounding
    assert scaled_err(lm3[:, 0::2], lm1[:, 1::2]) <= 1e-4 and scaled_err(lm3[:, 1::2], lm1[:, 0::2]) <= 1e-4
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


@pytest.mark.parametrize("kw,T", [
    (dict(), 88200),                                             # stem-per-wave-pair kernel
    (dict(n_fft=2048, hop_length=512, n_mels=80), 66150),        # generic kernel, even/odd packed FFT
    (dict(n_fft=1024, hop_length=256, n_mels=256), 44100),       # generic kernel, 4 band slots
    (dict(n_fft=512, hop_length=128, n_mels=64), 30001),         # odd length: scalar loads
])
def test_pcm16_input_is_bit_identical_to_fp32_of_the_same_samples(kw, T):
    """int16 PCM ingest: the kernels convert s * 2^-15 exactly, so features and log-mel must EQUAL the
    fp32 entry point run on float(pcm) / 32768 -- packed (B,8,T) tensor and the collate-style dict of views."""
    from mst_amd import ingest
    x = torch.stack([cases.synth_clip(c, T) for c in (2, 4)], 0)
    q = ingest.float_to_pcm16(x)
    xf = q.float() / 32768.0
    ext = fe(**kw)
    f0, lm0 = run(xf, ext)
    lm1, f1 = ext.plan().forward(q.cuda())
    assert torch.equal(f1.cpu(), f0) and torch.equal(lm1.cpu(), lm0)
    f2, lm2 = ext.features_and_logmel(ingest.stems_views(q.cuda()))
    assert torch.equal(f2.cpu(), f0) and torch.equal(lm2.cpu(), lm0)
    # and against the oracle on the dequantised samples
    rf, rmel = ofeat.extract_all_features(xf, kw.get("sample_rate", 44100), kw.get("n_fft", 1024),
                                          kw.get("hop_length", 256), kw.get("n_mels", 128), return_mel=True)
    check_feats(f1.cpu(), rf)


def test_device_stager_overlapped_h2d():
    """Pinned double-buffered staging: every staged batch arrives intact while the previous one is being consumed."""
    from mst_amd import ingest
    ext = fe()
    T, N = 44100, 3
    batches = [ingest.float_to_pcm16(torch.stack([cases.synth_clip(10 * b + c, T) for c in range(N)], 0))
               for b in range(5)]
    st = ingest.DeviceStager((N, 8, T), torch.int16, "cuda")
    fut = st.submit(batches[0])
    outs = []
    for b in range(5):
        x = fut.get()
        nxt = st.submit(batches[b + 1]) if b + 1 < 5 else None
        lm, f = ext.plan().forward(x)
        st.release(fut)
        outs.append((f.clone(), x.clone()))
        fut = nxt
    torch.cuda.synchronize()
    for b in range(5):
        assert torch.equal(outs[b][1].cpu(), batches[b])
        _, fr = ext.plan().forward(batches[b].cuda())
        assert torch.equal(outs[b][0], fr)


@pytest.mark.parametrize("T", [1024, 4096 + 256 * 3, 256 * 33, 256 * 55 + 128, 256 * 97 + 4, 256 * 1024 + 252])
def test_two_stage_a_kernels_agree_on_ragged_run_partitions(T):
    """The default stage-A kernel (sliding window, packed radix-16x8x8 FFT, segment mel) and the generic kernel (radix
    8x8x4x4 FFT, per-band gather) share no arithmetic and no frame / run / block bookkeeping.  Lengths chosen so that the
    number of frames is not a multiple of the block, of the run length, or of the hop (last partial frame), incl. one
    clip shorter than a run.  Kernel vs kernel: the 1e-4 log-mel bound (two different fp32 FFT factorisations differ by
    their rounding noise); both against the oracle with the noise-unit criterion of check_logmel."""
    x = torch.stack([cases.synth_clip(c, T) for c in (3, 8, 9)], 0)
    ext = fe()
    os.environ.pop("MST_MELFEAT_GENERIC", None)
    f_spw, lm_spw = run(x, ext)
    os.environ["MST_MELFEAT_GENERIC"] = "1"
    try:
        f_gen, lm_gen = run(x, ext)
    finally:
        os.environ.pop("MST_MELFEAT_GENERIC", None)
    assert scaled_err(lm_spw, lm_gen) <= 1e-4
    check_feats(f_spw, f_gen.numpy())
    rf, rmel = ofeat.extract_all_features(x, return_mel=True)
    check_logmel(lm_spw, torch.log(rmel + 1e-10), x=x)
    check_feats(f_spw, rf)
    # batch independence, bit for bit, for both kernels
    f1, lm1 = run(x[1:2], ext)
    assert torch.equal(f1[0], f_spw[1]) and torch.equal(lm1[0], lm_spw[1])


def test_real_music_log_mel_at_n_fft_2048_matches_pocketfft_noise():
    """The reference's working launcher configuration (scripts/train_baseline.sh: 2048 / 512 / 80 mels) on real music
    (the two 10 s crops of assets/song_A.wav).  Round 1's even/odd-packed real FFT carried 3-10x pocketfft's rounding noise
    in the top octave of low-passed stems; the 2048-point sliding-window kernel (two 1024-point FFTs + one DIT step on the
    L + i s R packing) has to meet the same measured-noise criterion as the 1024-point kernel, and the features 1e-4."""
    x = cases.song_a_clips()
    ext = fe(n_fft=2048, hop_length=512, n_mels=80)
    f, lm = run(x, ext)
    assert tuple(lm.shape) == (2, 8, 80, 1 + 441000 // 512)
    rf, rmel = ofeat.extract_all_features(x, 44100, 2048, 512, 80, return_mel=True)
    check_logmel(lm, torch.log(rmel + 1e-10), x=x, cfg=(44100, 2048, 512, 80))
    check_feats(f, rf)
    # batch independence and the generic kernel as a second opinion on bookkeeping (ragged run partition at 862 frames)
    f1, lm1 = run(x[1:2], ext)
    assert torch.equal(f1[0], f[1]) and torch.equal(lm1[0], lm[1])
    os.environ["MST_MELFEAT_GENERIC"] = "1"
    try:
        fg, lmg = run(x, ext)
    finally:
        os.environ.pop("MST_MELFEAT_GENERIC", None)
    check_feats(f, fg.numpy())
