"""Pin the CPU oracle against golden vectors produced by the reference's own code
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import augment as oaug
from oracle import encoder as oenc
from oracle import features as ofeat
from oracle import loss as oloss
from oracle import mel as omel

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def close(a, b, rtol=1e-4, atol=1e-5):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_fbanks_bit_exact():
    g = load("fbanks.npz")
    for n_fft, n_mels in ((1024, 128), (1024, 256), (2048, 80)):
        fb = omel.htk_fbank(44100, n_fft, n_mels).numpy()
        assert np.array_equal(fb, g[f"fb_{n_fft}_{n_mels}"])
        assert np.array_equal(omel.hann_periodic(n_fft).numpy(), g[f"win_{n_fft}"])
    fb = g["fb_1024_128"]
    assert (fb != 0).sum() == 1008 and (fb != 0).sum(1).max() <= 2 and not fb[:, 0].any()


@pytest.mark.parametrize("name", cases.FEATURE_CASES)
def test_features_edge_cases(name):
    g = load("features.npz")
    x = cases.feature_case(name, 44100)
    close(cases.checksum(x), g[f"{name}.in_checksum"], rtol=1e-12, atol=0)
    f = ofeat.extract_all_features(x[None])[0].numpy()
    assert f.shape == (64,)
    close(f, g[f"{name}.features"], rtol=1e-4, atol=2e-5)


def test_features_known_values():
    """Expected edge-case values listed in SURVEY 8c(iii)."""
    g = load("features.npz")
    f = g["silent_vocals.features"]
    v = f[49:64]  # vocals: dyn(6) rel(1) spec(5) stereo(3)
    assert v[0] == 0 and v[1] == 0 and v[2] == -100 and v[3] == -100      # rms, crest
    assert v[10] == 0 and abs(v[11] - 1.0) < 1e-5                          # tilt, flatness
    assert v[12] == -100 and v[13] == 0                                    # ILD, corr
    m = g["mono.features"]
    assert abs(m[12]) < 1e-5 and abs(m[13] - 1.0) < 1e-5 and abs(m[14]) < 1e-9  # bass ILD 0, corr 1, MSR 0
    w = g["white.features"]
    assert abs(w[0] - 0.1) < 2e-3 and abs(w[4] + 20.69) < 0.05 and abs(w[6] + 6.02) < 0.05


def test_features_full_size_cfg2_detailed_odd():
    g = load("features.npz")
    for c in (0, 1):
        x = cases.synth_clip(c, 441000)
        close(cases.checksum(x), g[f"synth10s_{c}.in_checksum"], rtol=1e-12, atol=0)
        close(ofeat.extract_all_features(x[None])[0].numpy(), g[f"synth10s_{c}.features"], atol=2e-5)
    x = cases.feature_case("synth1", 66150)
    close(ofeat.extract_all_features(x[None], 44100, 2048, 512, 80)[0].numpy(), g["cfg2.features"], atol=2e-5)
    x = cases.feature_case("synth", 44100)
    f = ofeat.extract_all_features(x[None], detailed=True, n_bins=32)[0].numpy()
    assert f.shape == (180,)
    close(f, g["detailed.features"], atol=2e-5)
    x = cases.feature_case("synth1", 30001)
    close(ofeat.extract_all_features(x[None])[0].numpy(), g["odd.features"], atol=2e-5)


def test_logmel():
    g = load("logmel.npz")
    for name, T in (("synth", 4096), ("synth1", 22050), ("one_sided", 8192)):
        x = cases.feature_case(name, T)[None]
        lm = omel.logmel(x).numpy()
        assert lm.shape == (1, 8, 128, 1 + T // 256)
        close(lm, g[f"{name}_{T}.logmel"], rtol=1e-5, atol=1e-5)
    lm = omel.logmel(cases.synth_clip(0, 441000)[None])[0]
    assert lm.shape == (8, 128, 1723)
    close(lm.flatten()[torch.from_numpy(g["synth10s_0.logmel_idx"])].numpy(), g["synth10s_0.logmel_samples"],
          rtol=1e-5, atol=1e-5)
    close(lm.double().sum(-1).numpy(), g["synth10s_0.logmel_rowsum"], rtol=1e-6, atol=1e-3)
    lm2 = omel.logmel(cases.feature_case("synth1", 66150)[None], 44100, 2048, 512, 80).numpy()
    close(lm2, g["cfg2.logmel"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag,cfg,T", [("default", cases.CFG_DEFAULT, 441000),
                                        ("cfg2", cases.CFG_BASELINE_SH, 441000),
                                        ("default_short", cases.CFG_DEFAULT, 66150)])
def test_encoder(tag, cfg, T):
    g = load("encoder.npz")
    sd = cases.make_state_dict(cfg, seed=42)
    x = torch.stack([cases.synth_clip(c, T) for c in (0, 1)], 0)
    close(cases.checksum(x), g[f"{tag}.in_checksum"], rtol=1e-12, atol=0)
    feats = ofeat.extract_all_features(x, cfg["sample_rate"], cfg["n_fft"], cfg["hop_length"], cfg["n_mels"])
    close(feats.numpy(), g[f"{tag}.features"], atol=2e-5)
    taps = {}
    emb = oenc.encoder_forward(sd, x, torch.from_numpy(g[f"{tag}.features"]), cfg["sample_rate"], cfg["n_fft"],
                               cfg["hop_length"], cfg["n_mels"], cfg["split_size"], cfg["overlap"], taps)
    assert tuple(emb.shape) == (2, cfg["embed_dim"])
    close(taps["film"].numpy(), g[f"{tag}.film"], rtol=1e-5, atol=1e-6)
    pin = taps["pool_in"]
    assert tuple(pin.shape) == tuple(g[f"{tag}.pool_in_shape"])
    close(pin.flatten()[torch.from_numpy(g[f"{tag}.pool_in_idx"])].numpy(), g[f"{tag}.pool_in_samples"],
          rtol=1e-4, atol=1e-5)
    ns = cases.n_subbands(cfg["n_mels"], cfg["split_size"], cfg["overlap"])
    for i in (0, ns // 2, ns - 1):
        p1 = taps[f"pool1_{i}"]
        assert tuple(p1.shape) == tuple(g[f"{tag}.pool1_{i}_shape"])
        close(p1.flatten()[torch.from_numpy(g[f"{tag}.pool1_{i}_idx"])].numpy(), g[f"{tag}.pool1_{i}_samples"],
              rtol=1e-4, atol=1e-5)
    scale = np.abs(g[f"{tag}.embedding"]).max()
    close(emb.numpy(), g[f"{tag}.embedding"], rtol=1e-4, atol=1e-5 * scale)


def test_state_dict_keys_match_reference():
    g = load("encoder.npz")
    for tag, cfg in (("default", cases.CFG_DEFAULT), ("cfg2", cases.CFG_BASELINE_SH)):
        ref_keys = set(g[f"{tag}.state_dict_keys"].tolist())
        mine = set(cases.state_dict_shapes(cfg))
        extra = ref_keys - mine
        assert mine <= ref_keys
        assert extra == {"audio_encoder.mel_preprocessor.mel_transform.spectrogram.window",
                         "audio_encoder.mel_preprocessor.mel_transform.mel_scale.fb"}


def test_infonce():
    g = load("infonce.npz")
    gen = torch.Generator().manual_seed(5)
    for tag, n, d, nsong in (("pairs48", 48, 768, 24), ("gathered384", 384, 768, 192), ("triples", 12, 16, 4)):
        e = torch.randn(n, d, generator=gen)
        close(cases.checksum(e), g[f"{tag}.emb_checksum"], rtol=1e-12, atol=0)
        close(oloss.info_nce(e, torch.arange(n) % nsong, 0.1).item(), g[f"{tag}.loss"], rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        oloss.info_nce(torch.randn(4, 8), torch.arange(4))


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 11])
def test_augment_trace_and_audio(seed):
    g = load("augment.npz")
    x = cases.feature_case("synth1", 33075)
    real_rand = torch.rand
    draws = []

    def rand(*a, **k):
        v = real_rand(*a, **k)
        draws.append(float(v.flatten()[0]))
        return v

    torch.manual_seed(seed)
    torch.rand = rand
    try:
        y, trace = oaug.augment_stems(omel.tensor_to_stems_dict(x))
    finally:
        torch.rand = real_rand
    # RNG consumption (decisions) bit-exact
    assert np.array_equal(np.array(draws, dtype=np.float64), g[f"seed{seed}.rand_draws"])
    assert int(g[f"seed{seed}.reverb"]) == int("reverb_ir" in trace)
    y8 = omel.stems_dict_to_tensor(y)
    idx = torch.from_numpy(g[f"seed{seed}.idx"])
    close(y8.flatten()[idx].numpy(), g[f"seed{seed}.samples"], rtol=1e-5, atol=1e-6)
    close(y8.double().sum(-1).numpy(), g[f"seed{seed}.chan_sum"], rtol=1e-5, atol=1e-3)
    close((y8.double() ** 2).sum(-1).numpy(), g[f"seed{seed}.chan_sqsum"], rtol=1e-5, atol=1e-6)


def test_augment_single_effects():
    g = load("augment.npz")
    x = cases.feature_case("synth1", 33075)[4:6]
    close(oaug.compress(x).numpy()[:, :4096], g["compress.samples"], rtol=1e-6, atol=1e-8)
    # non-default threshold / ratio (the reference's apply_compression is parametric): its own fixture
    gp = np.load(os.path.join(G, "augment_compress_param.npz"))
    xp = x.clone()
    xp[:, :64] = 0.0
    xp[:, 64:128] *= 8.0
    for tag in ("m12_2", "m30_8", "m6_1p5"):
        thr, ratio = (float(v) for v in gp[f"{tag}.params"])
        close(oaug.compress(xp, thr, ratio).numpy()[:, :8192], gp[f"{tag}.samples"], rtol=1e-6, atol=1e-8)
    torch.manual_seed(123)
    close(oaug.reverb(x, oaug.make_ir(44100)).numpy()[:, :4096], g["reverb.out_head"], rtol=1e-5, atol=1e-6)


def test_song_a_real_music_config0():
    """BASELINE configs[0]: real music (two 10 s crops of the reference's assets/song_A.wav, pseudo-separated)."""
    g = load("song_a.npz")
    x = cases.song_a_clips()
    assert tuple(x.shape) == (2, 8, 441000)
    close(cases.checksum(x), g["in_checksum"], rtol=1e-12, atol=0)
    close(ofeat.extract_all_features(x).numpy(), g["features"], atol=2e-5)
    lm = omel.logmel(x)
    close(lm.flatten()[torch.from_numpy(g["logmel_idx"])].numpy(), g["logmel_samples"], rtol=1e-5, atol=1e-5)
    sd = cases.make_state_dict(cases.CFG_DEFAULT, seed=42)
    emb = oenc.encoder_from_logmel(sd, lm, torch.from_numpy(g["features"]))
    close(emb.numpy(), g["embedding"], rtol=1e-4, atol=1e-5 * np.abs(g["embedding"]).max())


def test_f16_training_oracle_is_pinned_to_torch_autocast():
    """The f16 training mode's arithmetic contract (oracle/train_f16.py: conv operands rounded to float16, fp32 accumulation, the
    output stored as float16) is what the reference's `--use_amp` step runs -- `torch.autocast(float16)` around the model
    (src/train.py:251-253).  PyTorch's own autocast convolution (CPU build, same semantics as the GPU one) pins the restatement:
    with the bias rounded too, as autocast does, at least 99 % of the outputs are identical float16 values and the rest are the
    neighbouring float16 (accumulation order); with the bias kept in fp32, as the kernels do -- it cancels in the batch-statistics
    BatchNorm that follows -- every output is within one rounding of y plus the rounding of the bias.  The input gradient of the oracle equals autograd's through the
    same rounded operands with the upstream gradient rounded to float16."""
    import torch.nn.functional as F
    from oracle.encoder import round_f16_ideal as r
    from oracle.train_f16 import _F16OperandConv
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 8, 20, 40, generator=g)
    w = torch.randn(32, 8, 7, 7, generator=g) * 0.05
    b = torch.randn(32, generator=g) * 0.1
    with torch.autocast("cpu", dtype=torch.float16):
        y = F.conv2d(x, w, b, padding=3)
    assert y.dtype == torch.float16
    y = y.double()
    ulp = torch.ldexp(torch.ones_like(y), torch.frexp(y)[1] - 11)           # float16 spacing at |y|
    ours = _F16OperandConv.apply(x.double(), w.double(), b.double(), (3, 3))
    assert ((ours - y).abs() <= 2.0 ** -10 * (y.abs() + b.abs().max())).all()   # the roundings of y and of autocast's bias
    amp_like = r(F.conv2d(r(x).double(), r(w).double(), r(b).double(), padding=3))
    same = (amp_like == y).double().mean().item()
    # the rest: a neighbouring float16 where PyTorch's fp32 accumulation order rounds the other way (a few ulps of y where the
    # products cancel: the error is relative to the partial sums, not to y)
    assert same >= 0.99 and ((amp_like - y).abs() <= torch.maximum(ulp * 1.0001, torch.full_like(y, 2.0 ** -12))).all(), same
    # gradients: dx = conv^T(r(dy), r(w)), dW = corr(r(x), r(dy)); the output rounding is transparent (a cast)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    dy = torch.randn(2, 32, 20, 40, generator=g).double() * 1e-3
    _F16OperandConv.apply(xd, wd, b.double(), (3, 3)).backward(dy)
    xr, wr = r(x.double()).requires_grad_(True), r(w.double()).requires_grad_(True)
    F.conv2d(xr, wr, b.double(), padding=3).backward(r(dy))
    close(xd.grad.numpy(), xr.grad.numpy(), rtol=1e-12, atol=1e-15)
    close(wd.grad.numpy(), wr.grad.numpy(), rtol=1e-12, atol=1e-15)


CFG_C5 = dict(sample_rate=44100, n_fft=1024, hop_length=256, n_mels=256, split_size=20, overlap=10, embed_dim=768)


@pytest.mark.parametrize("tag", ["default", "c5"])
@pytest.mark.parametrize("lossname", ["infonce", "proj"])
def test_training_oracle_is_pinned_by_the_reference_training_step(tag, lossname):
    """tests/golden/train.npz holds what the REFERENCE's modules compute in training mode (src/train.py:246-262,292-296;
    BatchNorm with batch statistics, Dropout p = 0): loss, embeddings, batch statistics and sampled gradients of every
    parameter, in fp32 and float64.  The oracle's train-mode path (bn_training=True + autograd) evaluated in float64 must
    reproduce the float64 fixture to 1e-9, and in fp32 the fp32 fixture to fp32 rounding: from here on the GPU tests may
    use the oracle (and these fixtures directly) as the yardstick of the training kernels."""
    g = load("train.npz")
    cfg = cases.CFG_DEFAULT if tag == "default" else CFG_C5
    B, T = 4, 66150
    x = cases.pcm_batch(B, T)
    assert cases.checksum(x) == g[f"{tag}.in_checksum"].tolist()   # integer-built input: exact on every machine
    feats = torch.from_numpy(g[f"{tag}.features"])
    labels = torch.from_numpy(g[f"{tag}.labels"])
    R = torch.randn(B, cfg["embed_dim"], generator=torch.Generator().manual_seed(77))
    assert np.allclose(cases.checksum(R), g[f"{tag}.R_checksum"], rtol=1e-12)
    names = [str(n) for n in g[f"{tag}.param_names"]]
    for prec, dt, tol in (("f64", torch.float64, 1e-9), ("f32", torch.float32, 2e-4)):
        sd = {k: (v.to(dt).requires_grad_(True) if v.dtype.is_floating_point else v)
              for k, v in cases.make_state_dict(cfg, seed=42).items()}
        lm = omel.logmel(x.to(dt), cfg["sample_rate"], cfg["n_fft"], cfg["hop_length"], cfg["n_mels"])
        taps = {}
        emb = oenc.encoder_from_logmel(sd, lm, feats.to(dt), cfg["split_size"], cfg["overlap"], taps=taps, bn_training=True)
        loss = oloss.info_nce(emb, labels, 0.1) if lossname == "infonce" else (emb * R.to(dt)).sum()
        loss.backward()
        close(loss.item(), g[f"{tag}.{prec}.{lossname}.loss"], rtol=max(tol, 1e-6), atol=0)
        close(emb.detach(), g[f"{tag}.{prec}.embedding"], rtol=tol, atol=tol * float(np.abs(g[f"{tag}.{prec}.embedding"]).max()))
        ns = oenc.n_subbands(cfg["n_mels"], cfg["split_size"], cfg["overlap"])
        for i in (0, ns // 2, ns - 1):
            for bn in ("bn1", "bn2"):
                mean, var = taps[f"{bn}_{i}"]
                close(mean.detach(), g[f"{tag}.{prec}.{bn}_{i}.batch_mean"], rtol=tol, atol=tol)
                close(var.detach(), g[f"{tag}.{prec}.{bn}_{i}.batch_var"], rtol=tol, atol=tol)
        worst = (0.0, "")
        for j, n in enumerate(names):
            ref = g[f"{tag}.{prec}.{lossname}.grad_samples.{n}"].astype(np.float64)
            gmax = float(g[f"{tag}.{prec}.{lossname}.grad_norm.{n}"][1])
            got = sd[n].grad.double().flatten()[cases.sample_idx(sd[n].numel(), 512, 1000 + j)].numpy()
            if gmax > 0:
                worst = max(worst, (float(np.abs(got - ref).max() / gmax), n))
        # fp32: two fp32 evaluations of the same graph (functional ops here, nn.Modules there) differ by rounding in the
        # cancellation-prone conv1 weight gradients; float64: identical arithmetic
        assert worst[0] <= (1e-8 if prec == "f64" else 5e-3), worst
