"""Generate golden vectors by running the REFERENCE's own Python code in the build container.

Usage (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_golden.py

Imports /root/reference/src/{mixing_utils,model,loss}.py unmodified, with the stand-in
`torchaudio` of oracle/torchaudio_standin on sys.path (torchaudio is not installed here).
Writes small .npz fixtures next to this file.  Inputs are NOT stored: tests rebuild them
from tests/cases.py seeds and verify the stored input checksums.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "torchaudio_standin"))
sys.path.insert(0, "/root/reference/src")

import cases  # noqa: E402
import mixing_utils as ref_mu  # noqa: E402  (reference)
import model as ref_model  # noqa: E402      (reference)
import loss as ref_loss  # noqa: E402        (reference)

torch.set_num_threads(8)


def stems_dict(x):
    return {s: x[..., 2 * i:2 * i + 2, :] for i, s in enumerate(cases.STEMS)}


def sample_idx(n, k=2048, seed=0):
    return np.random.default_rng(seed).choice(n, size=min(k, n), replace=False).astype(np.int64)


def gen_fbanks():
    out = {}
    for sr, n_fft, n_mels in ((44100, 1024, 128), (44100, 1024, 256), (44100, 2048, 80)):
        mt = ref_mu.MixingFeatureExtractor(sr, n_fft, n_fft // 4, n_mels).mel_transform
        out[f"fb_{n_fft}_{n_mels}"] = mt.mel_scale.fb.numpy()
        out[f"win_{n_fft}"] = mt.spectrogram.window.numpy()
    np.savez_compressed(os.path.join(HERE, "fbanks.npz"), **out)


def gen_features():
    out = {}
    fe = ref_mu.MixingFeatureExtractor()
    for name in cases.FEATURE_CASES:
        x = cases.feature_case(name, 44100)
        out[f"{name}.in_checksum"] = np.array(cases.checksum(x))
        out[f"{name}.features"] = fe.extract_all_features(stems_dict(x)).numpy()
    # full-size clips (10 s) incl. per-piece outputs
    for c in (0, 1):
        x = cases.synth_clip(c, 441000)
        out[f"synth10s_{c}.in_checksum"] = np.array(cases.checksum(x))
        out[f"synth10s_{c}.features"] = fe.extract_all_features(stems_dict(x)).numpy()
    # second config in real use (train_baseline.sh) and detailed-spectral mode
    fe2 = ref_mu.MixingFeatureExtractor(44100, 2048, 512, 80)
    x = cases.feature_case("synth1", 66150)
    out["cfg2.in_checksum"] = np.array(cases.checksum(x))
    out["cfg2.features"] = fe2.extract_all_features(stems_dict(x)).numpy()
    fe3 = ref_mu.MixingFeatureExtractor(use_detailed_spectral=True, n_spectral_bins=32)
    out["detailed.features"] = fe3.extract_all_features(stems_dict(cases.feature_case("synth", 44100))).numpy()
    assert fe3.get_feature_dim() == out["detailed.features"].shape[0] == 180
    # odd length / not a hop multiple
    x = cases.feature_case("synth1", 30001)
    out["odd.in_checksum"] = np.array(cases.checksum(x))
    out["odd.features"] = fe.extract_all_features(stems_dict(x)).numpy()
    np.savez_compressed(os.path.join(HERE, "features.npz"), **out)


def gen_logmel():
    out = {}
    pre = ref_model.MelSpectrogramPreprocessor()
    for name, T in (("synth", 4096), ("synth1", 22050), ("one_sided", 8192)):
        x = cases.feature_case(name, T)[None]
        out[f"{name}_{T}.in_checksum"] = np.array(cases.checksum(x))
        out[f"{name}_{T}.logmel"] = pre(stems_dict(x)).numpy()
    x = cases.synth_clip(0, 441000)[None]
    lm = pre(stems_dict(x))[0]  # (8,128,1723)
    idx = sample_idx(lm.numel(), 4096, 1)
    out["synth10s_0.logmel_rowsum"] = lm.double().sum(-1).numpy()
    out["synth10s_0.logmel_idx"] = idx
    out["synth10s_0.logmel_samples"] = lm.flatten()[idx].numpy()
    pre2 = ref_model.MelSpectrogramPreprocessor(44100, 2048, 512, 80)
    x = cases.feature_case("synth1", 66150)[None]
    out["cfg2.logmel"] = pre2(stems_dict(x)).numpy()
    np.savez_compressed(os.path.join(HERE, "logmel.npz"), **out)


def gen_encoder():
    out = {}
    for tag, cfg, T in (("default", cases.CFG_DEFAULT, 441000), ("cfg2", cases.CFG_BASELINE_SH, 441000),
                        ("default_short", cases.CFG_DEFAULT, 66150)):
        m = ref_model.MixingStyleEncoder(channels=8, feature_dim=64, **cfg).eval()
        sd = cases.make_state_dict(cfg, seed=42)
        full = dict(m.state_dict())
        for k, v in sd.items():
            assert full[k].shape == v.shape, (k, full[k].shape, v.shape)
        missing = set(full) - set(sd)
        assert all("mel_transform" in k for k in missing), missing
        full.update(sd)
        m.load_state_dict(full, strict=True)
        out[f"{tag}.state_dict_keys"] = np.array(sorted(m.state_dict().keys()))
        x = torch.stack([cases.synth_clip(c, T) for c in (0, 1)], 0)
        fe = ref_mu.MixingFeatureExtractor(cfg["sample_rate"], cfg["n_fft"], cfg["hop_length"], cfg["n_mels"])
        feats = torch.stack([fe.extract_all_features(stems_dict(x[b])) for b in range(2)], 0)
        taps = {}
        hooks = [m.audio_encoder.attention_pooling.register_forward_hook(
            lambda mod, inp, o: taps.__setitem__("pool_in", inp[0].detach()))]
        for i, cnn in enumerate(m.audio_encoder.subnet_cnns):
            hooks.append(cnn.pool1.register_forward_hook(
                lambda mod, inp, o, i=i: taps.__setitem__(f"pool1_{i}", o.detach())))
        with torch.no_grad():
            film = m.film_encoder(feats)
            emb = m(stems_dict(x), feats)
        for h in hooks:
            h.remove()
        ns = m.audio_encoder.n_subbands
        out[f"{tag}.in_checksum"] = np.array(cases.checksum(x))
        out[f"{tag}.features"] = feats.numpy()
        out[f"{tag}.film"] = torch.cat([torch.cat([film[f"gamma1_{i}"], film[f"beta1_{i}"], film[f"gamma2_{i}"],
                                                   film[f"beta2_{i}"]], 1) for i in range(ns)], 1).numpy()
        out[f"{tag}.embedding"] = emb.numpy()
        pin = taps["pool_in"]
        out[f"{tag}.pool_in_shape"] = np.array(pin.shape)
        idx = sample_idx(pin.numel(), 4096, 2)
        out[f"{tag}.pool_in_idx"] = idx
        out[f"{tag}.pool_in_samples"] = pin.flatten()[idx].numpy()
        out[f"{tag}.pool_in_rowsum"] = pin.double().sum(-1).numpy()
        for i in (0, ns // 2, ns - 1):
            p1 = taps[f"pool1_{i}"]
            idx = sample_idx(p1.numel(), 2048, 3 + i)
            out[f"{tag}.pool1_{i}_shape"] = np.array(p1.shape)
            out[f"{tag}.pool1_{i}_idx"] = idx
            out[f"{tag}.pool1_{i}_samples"] = p1.flatten()[idx].numpy()
    np.savez_compressed(os.path.join(HERE, "encoder.npz"), **out)


def gen_infonce():
    out = {}
    crit = ref_loss.InfoNCELoss(temperature=0.1)
    g = torch.Generator().manual_seed(5)
    for tag, n, d, nsong in (("pairs48", 48, 768, 24), ("gathered384", 384, 768, 192), ("triples", 12, 16, 4)):
        e = torch.randn(n, d, generator=g)
        lab = torch.arange(n) % nsong
        out[f"{tag}.emb_checksum"] = np.array(cases.checksum(e))
        out[f"{tag}.loss"] = np.array(crit(e, lab).item())
        out[f"{tag}.seed_note"] = np.array("generator seed 5, drawn in order pairs48, gathered384, triples")
    np.savez_compressed(os.path.join(HERE, "infonce.npz"), **out)


def gen_augment():
    """Decision trace is captured by logging torch.rand / torch.randn draws made by the reference."""
    out = {}
    aug = ref_mu.AudioAugmenter(sample_rate=44100, gain_range=9.0, prob=0.5)
    real_rand, real_randn = torch.rand, torch.randn
    for seed in (0, 1, 2, 3, 4, 5, 11):
        x = cases.feature_case("synth1", 33075)
        draws = []

        def rand(*a, **k):
            v = real_rand(*a, **k)
            draws.append(float(v.flatten()[0]))
            return v

        ir_box = []

        def randn(*a, **k):
            v = real_randn(*a, **k)
            ir_box.append(v.clone())
            return v

        torch.manual_seed(seed)
        torch.rand, torch.randn = rand, randn
        try:
            y = aug.augment_stems(stems_dict(x))
        finally:
            torch.rand, torch.randn = real_rand, real_randn
        y8 = torch.cat([y[s] for s in cases.STEMS], 0)
        out[f"seed{seed}.rand_draws"] = np.array(draws, dtype=np.float64)
        out[f"seed{seed}.reverb"] = np.array(len(ir_box))
        if ir_box:
            out[f"seed{seed}.ir_randn_head"] = ir_box[0][:8].numpy()
            out[f"seed{seed}.ir_randn_sum"] = np.array(float(ir_box[0].double().sum()))
        idx = sample_idx(y8.numel(), 4096, 100 + seed)
        out[f"seed{seed}.idx"] = idx
        out[f"seed{seed}.samples"] = y8.flatten()[idx].numpy()
        out[f"seed{seed}.chan_sum"] = y8.double().sum(-1).numpy()
        out[f"seed{seed}.chan_sqsum"] = (y8.double() ** 2).sum(-1).numpy()
    # individual effects, deterministic
    x = cases.feature_case("synth1", 33075)[4:6]
    out["compress.samples"] = aug.apply_compression(x).numpy()[:, :4096]
    torch.manual_seed(123)
    out["reverb.out_head"] = aug.apply_reverb(x).numpy()[:, :4096]
    torch.manual_seed(9)   # rand < 0.5 ? high : low
    first = float(real_rand(1)); torch.manual_seed(9)
    out["tilt.coin"] = np.array(first)
    out["tilt.out_head"] = aug.apply_spectral_tilt(x).numpy()[:, :4096]
    torch.manual_seed(10)
    out["bw.out_head"] = aug.apply_bandwidth_limit(x).numpy()[:, :4096]
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)


def gen_song_a():
    """BASELINE configs[0]: two 10 s crops of assets/song_A.wav (the reference's own asset; data, stored as int16),
    fixed pseudo-separation in place of SCNet, reference features + log-mel + embeddings."""
    import wave
    with wave.open("/root/reference/assets/song_A.wav", "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (2, 2, 44100)
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, 2).T.copy()
    crops = {"crop0": pcm[:, 0:441000], "crop1": pcm[:, 220538:220538 + 441000]}
    np.savez_compressed(os.path.join(HERE, "song_a_crops.npz"), **crops)
    x = cases.song_a_clips()
    out = {"in_checksum": np.array(cases.checksum(x))}
    fe = ref_mu.MixingFeatureExtractor()
    feats = torch.stack([fe.extract_all_features(stems_dict(x[b])) for b in range(2)], 0)
    out["features"] = feats.numpy()
    cfg = cases.CFG_DEFAULT
    m = ref_model.MixingStyleEncoder(channels=8, feature_dim=64, **cfg).eval()
    full = dict(m.state_dict())
    full.update(cases.make_state_dict(cfg, seed=42))
    m.load_state_dict(full, strict=True)
    with torch.no_grad():
        lm = m.audio_encoder.mel_preprocessor(stems_dict(x))
        emb = m(stems_dict(x), feats)
    idx = sample_idx(lm.numel(), 8192, 7)
    out["logmel_idx"], out["logmel_samples"] = idx, lm.flatten()[idx].numpy()
    out["logmel_rowsum"] = lm.double().sum(-1).numpy()
    out["embedding"] = emb.numpy()
    np.savez_compressed(os.path.join(HERE, "song_a.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["fbanks", "features", "logmel", "encoder", "infonce", "augment", "song_a"]
    for w in which:
        print("generating", w, flush=True)
        globals()["gen_" + w]()
    print("done")
