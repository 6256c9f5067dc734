"""Generate golden vectors by running the REFERENCE's own Python code in the build container.

Usage (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_golden.py

Imports /root/reference/src/{mixing_utils,model,loss,data,validation_utils}.py unmodified, with the stand-in
`torchaudio` of oracle/torchaudio_standin on sys.path (torchaudio is not installed here) and, for data.py /
validation_utils.py, the empty-shell `librosa` / SCNet `utils.*` modules of oracle/ref_import_standins.
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
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_import_standins"))   # librosa, SCNet `utils.*` shells (README there)
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


def gen_compress_param():
    """`apply_compression(audio, threshold, ratio)` of the reference (src/mixing_utils.py:435-447) at NON-default settings -- a
    separate small fixture (the other fixtures are not regenerated): one mild and one hard setting, on a signal with samples on both
    sides of either threshold, exact zeros and both signs."""
    out = {}
    aug = ref_mu.AudioAugmenter(sample_rate=44100, gain_range=9.0, prob=0.5)
    x = cases.feature_case("synth1", 33075)[4:6].clone()
    x[:, :64] = 0.0
    x[:, 64:128] *= 8.0        # loud samples, well above -6 dB
    out["input_checksum"] = np.array(float(x.double().abs().sum()))
    for tag, thr, ratio in (("m12_2", -12.0, 2.0), ("m30_8", -30.0, 8.0), ("m6_1p5", -6.0, 1.5)):
        out[f"{tag}.params"] = np.array([thr, ratio])
        out[f"{tag}.samples"] = aug.apply_compression(x, threshold=thr, ratio=ratio).numpy()[:, :8192]
    np.savez_compressed(os.path.join(HERE, "augment_compress_param.npz"), **out)


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


def gen_dataset():
    """SURVEY 8 a2-a4 + f3 from the reference's own src/data.py and src/validation_utils.py: numpy-RNG draw traces, clip
    checksums, features and collated batches of FMABaselineDataset / StyleTransferDataset on the toy tracks of
    tests/cases.py, retrieval metrics on seeded embeddings, and an embedding cache built by the reference loop."""
    import tempfile
    import data as ref_data                      # reference src/data.py
    import validation_utils as ref_vu            # reference src/validation_utils.py
    out = {}
    names = sorted(cases.TOY_TRACKS)
    cat8 = lambda d: torch.cat([d[s] for s in cases.STEMS], 0)  # noqa: E731
    with tempfile.TemporaryDirectory() as root:
        cases.write_toy_tracks(root)
        # --- FMABaselineDataset: crop starts, clips, features, collate (src/data.py:201-328)
        for nseg in (1, 2):
            ds = ref_data.FMABaselineDataset(root, clip_duration=0.25, num_segments=nseg)
            assert ds.clip_samples == cases.TOY_CLIP and len(ds) == len(names)
            where = {os.path.basename(d): i for i, d in enumerate(ds.track_dirs)}
            np.random.seed(42)
            with cases.RandintLog() as log:
                items = [ds[where[n]] for n in names]
            out[f"fma{nseg}.randint"] = np.array(log.calls, dtype=np.int64).reshape(-1, 3)
            out[f"fma{nseg}.clips_per_item"] = np.array([len(it[0]) for it in items])
            out[f"fma{nseg}.clip_checksum"] = np.array([cases.checksum(cat8(c)) for it in items for c in it[0]])
            out[f"fma{nseg}.clip_head"] = np.stack([cat8(c)[:, :4].numpy() for it in items for c in it[0]])
            out[f"fma{nseg}.features"] = np.stack([f.numpy() for it in items for f in it[1]])
            out[f"fma{nseg}.item_track"] = np.array([os.path.basename(it[3]) for it in items])
            assert all(it[2] == where[os.path.basename(it[3])] for it in items)
            sd, feats, labels, dirs = ref_data.baseline_collate_fn(items)
            out[f"fma{nseg}.collate_feature_shape"] = np.array(feats.shape)
            out[f"fma{nseg}.collate_label_track"] = np.array([names[[where[n] for n in names].index(int(l))] for l in labels])
            out[f"fma{nseg}.collate_dir_track"] = np.array([os.path.basename(d) for d in dirs])
            out[f"fma{nseg}.collate_checksum"] = np.array([cases.checksum(sd[s]) for s in cases.STEMS])
            out[f"fma{nseg}.collate_dtype"] = np.array([str(labels.dtype), str(feats.dtype), str(sd["bass"].dtype)])
            assert torch.equal(feats, torch.stack([f for it in items for f in it[1]]))
        # bad num_segments: the reference's error text
        try:
            ref_data.FMABaselineDataset(root, clip_duration=0.25, num_segments=3)[0]
        except ValueError as e:
            out["fma.bad_segments_error"] = np.array(str(e))
        # --- StyleTransferDataset draw order (src/data.py:467-538)
        st = ref_data.StyleTransferDataset(None, root, clip_duration=0.4)
        where = {os.path.basename(d): i for i, d in enumerate(st.track_dirs)}
        order = ["a_long", "c_lt2c", "e_mono", "d_ltc", "b_exact2c", "a_long"]
        np.random.seed(7)
        with cases.RandintLog() as log:
            items = [st[where[n]] for n in order]
        # the target-index draws depend on the directory listing order: store them as names
        out["st.order"] = np.array(order)
        out["st.listing"] = np.array([os.path.basename(d) for d in st.track_dirs])
        out["st.randint"] = np.array(log.calls, dtype=np.int64).reshape(-1, 3)
        out["st.input_checksum"] = np.array([cases.checksum(cat8(it[0])) for it in items])
        out["st.target_checksum"] = np.array([cases.checksum(cat8(it[1])) for it in items])
        out["st.target_features"] = np.stack([it[2].numpy() for it in items])
        i_sd, t_sd, tf = ref_data.style_transfer_collate_fn(items)
        out["st.collate_checksum"] = np.array([cases.checksum(i_sd[s]) + cases.checksum(t_sd[s]) for s in cases.STEMS])
        out["st.collate_feature_shape"] = np.array(tf.shape)
        # --- segment loading + embedding cache through the reference loop (src/validation_utils.py:15-74,106-214)
        seg = ref_vu.load_stems_segment(os.path.join(root, "c_lt2c"), 0.2, 0.25, 44100)   # runs past the end: zero padded
        out["vu.segment_shape"] = np.array(seg["drums"].shape)
        out["vu.segment_checksum"] = np.array(cases.checksum(torch.from_numpy(np.concatenate([seg[s] for s in cases.STEMS]))))
        seg = ref_vu.load_stems_segment(os.path.join(root, "e_mono"), 0.1, 0.25, 44100)    # mono stems -> stereo
        out["vu.segment_mono_checksum"] = np.array(cases.checksum(torch.from_numpy(np.concatenate([seg[s] for s in cases.STEMS]))))
        cfg = cases.CFG_DEFAULT
        m = ref_model.MixingStyleEncoder(channels=8, feature_dim=64, **cfg).eval()
        full = dict(m.state_dict())
        full.update(cases.make_state_dict(cfg, seed=42))
        m.load_state_dict(full, strict=True)
        ds = ref_data.FMABaselineDataset(root, clip_duration=0.25)
        where = {os.path.basename(d): i for i, d in enumerate(ds.track_dirs)}
        cache = ref_vu.build_embedding_cache(ds, [where[n] for n in names], m, ds.feature_extractor, None,
                                             torch.device("cpu"), query_duration=0.5)
        out["vu.cache_tracks"] = np.array([os.path.basename(p) for p in cache["track_paths"]])
        out["vu.cache_embeddings"] = cache["embeddings"].numpy()
        assert cache["track_indices"] == [where[n] for n in names]
    # --- retrieval metric on seeded embeddings (src/validation_utils.py:217-282)
    g = torch.Generator().manual_seed(3)
    pool = torch.randn(40, 32, generator=g)
    pool_idx = list(range(100, 140))
    q_idx = [100 + i for i in range(0, 40, 2)]
    queries = torch.stack([pool[i - 100] + 2.0 * torch.randn(32, generator=g) for i in q_idx])
    met = ref_vu.evaluate_retrieval_accuracy(queries, pool, q_idx, pool_idx, [1, 3, 5])
    out["ret.in_checksum"] = np.array(cases.checksum(pool) + cases.checksum(queries))
    out["ret.metrics"] = np.array([met["top_1_accuracy"], met["top_3_accuracy"], met["top_5_accuracy"]])
    ti, ts = ref_vu.retrieve_top_k(queries[0], pool, k=5)
    out["ret.top5_idx"], out["ret.top5_sim"] = ti.numpy(), ts.numpy()
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)


CFG_C5 = dict(sample_rate=44100, n_fft=1024, hop_length=256, n_mels=256, split_size=20, overlap=10, embed_dim=768)   # BASELINE configs[4] shapes


def gen_train():
    """The reference's own TRAINING arithmetic (src/train.py:246-262,292-296 on src/model.py in train mode): MixingStyleEncoder.train()
    -- BatchNorm with batch statistics (model.py:118,125), every nn.Dropout.p = 0 (dropout masks are not reproducible across
    implementations) -- followed by (a) InfoNCELoss(0.1) on labels [0, 0, 1, 1] and loss.backward(), the reference's step, and
    (b) the projection loss (embeddings * R).sum() with a fixed random R, which exercises every gradient path at O(1) magnitude
    (random-init embeddings of different clips are nearly parallel, so InfoNCE's gradient is small).  Stored per case: losses,
    train-mode embeddings, the batch statistics that bn1 / bn2 of three sub-bands saw, the running statistics after the step,
    and of EVERY parameter gradient 512 sampled entries (cases.sample_idx(numel, 512, 1000 + j), not stored) + its L2 norm, max
    and sum -- from the reference modules in fp32 and once more in float64 (the yardstick: two fp32 evaluations of a gradient
    differ by their own rounding)."""
    out = {}
    for tag, cfg, B, T in (("default", cases.CFG_DEFAULT, 4, 66150), ("c5", CFG_C5, 4, 66150)):
        x = cases.pcm_batch(B, T)   # integer-built PCM: bit-identical on the GPU box (see cases.pcm_clip)
        fe = ref_mu.MixingFeatureExtractor(cfg["sample_rate"], cfg["n_fft"], cfg["hop_length"], cfg["n_mels"])
        feats = torch.stack([fe.extract_all_features(stems_dict(x[b])) for b in range(B)], 0)
        labels = torch.arange(B) // 2
        R = torch.randn(B, cfg["embed_dim"], generator=torch.Generator().manual_seed(77))
        out[f"{tag}.in_checksum"] = np.array(cases.checksum(x))
        out[f"{tag}.features"] = feats.numpy()
        out[f"{tag}.labels"] = labels.numpy()
        out[f"{tag}.R_checksum"] = np.array(cases.checksum(R))
        for prec, dt in (("f32", torch.float32), ("f64", torch.float64)):
            for lossname in ("infonce", "proj"):
                torch.manual_seed(0)
                m = ref_model.MixingStyleEncoder(channels=8, feature_dim=64, **cfg)
                sd = cases.make_state_dict(cfg, seed=42)
                full = dict(m.state_dict())
                full.update(sd)
                m.load_state_dict(full, strict=True)
                for mod in m.modules():
                    if isinstance(mod, torch.nn.Dropout):
                        mod.p = 0.0
                m = m.to(dt).train()
                ns = m.audio_encoder.n_subbands
                stats, hooks = {}, []
                for i in (0, ns // 2, ns - 1):
                    cnn = m.audio_encoder.subnet_cnns[i]
                    for name, bn in (("bn1", cnn.bn1), ("bn2", cnn.bn2)):
                        hooks.append(bn.register_forward_hook(
                            lambda mod, inp, o, k=f"{name}_{i}": stats.__setitem__(k, (inp[0].detach().mean((0, 2, 3)),
                                                                                         inp[0].detach().var((0, 2, 3), unbiased=False)))))
                hooks.append(m.audio_encoder.attention_pooling.register_forward_hook(
                    lambda mod, inp, o: stats.__setitem__("pool_in", inp[0].detach())))
                gaps = {}
                if prec == "f64" and lossname == "infonce":
                    # how close every max-pool decision of the float64 run is to a tie: per sub-band the smallest gap between the
                    # two largest candidates of a window whose maximum is positive (ReLU active).  A gap below the fp32 noise of
                    # the candidates lets an fp32 evaluation route that window's gradient to the other position -- a step
                    # discontinuity of the gradient, 1/#windows of a (band, channel)'s tensors (~1e-2 for the second pooling)
                    def gap_hook(key):
                        def hook(mod, inp, o):
                            k = mod.kernel_size if isinstance(mod.kernel_size, tuple) else (mod.kernel_size,) * 2
                            u = torch.nn.functional.unfold(inp[0].detach().reshape(-1, 1, *inp[0].shape[2:]), k, stride=k)
                            top = u.topk(2, dim=1).values
                            gp = torch.where(top[:, 0] > 0, top[:, 0] - top[:, 1], torch.full_like(top[:, 0], float("inf")))
                            gaps[key] = float(gp.min())
                        return hook
                    for i, cnn in enumerate(m.audio_encoder.subnet_cnns):
                        hooks.append(cnn.pool1.register_forward_hook(gap_hook(("pool1", i))))
                        hooks.append(cnn.pool2.register_forward_hook(gap_hook(("pool2", i))))
                emb = m(stems_dict(x.to(dt)), feats.to(dt))                 # train.py:253 / :299
                if gaps:
                    out[f"{tag}.pool1_min_gap"] = np.array([gaps[("pool1", i)] for i in range(ns)])
                    out[f"{tag}.pool2_min_gap"] = np.array([gaps[("pool2", i)] for i in range(ns)])
                    print("   pool1 min gaps", " ".join(f"{v:.1e}" for v in out[f"{tag}.pool1_min_gap"]))
                    print("   pool2 min gaps", " ".join(f"{v:.1e}" for v in out[f"{tag}.pool2_min_gap"]), flush=True)
                if lossname == "infonce":
                    loss = ref_loss.InfoNCELoss(temperature=0.1)(emb, labels)   # train.py:256 / :302
                else:
                    loss = (emb * R.to(dt)).sum()
                loss.backward()                                              # train.py:292 / :323
                for h in hooks:
                    h.remove()
                out[f"{tag}.{prec}.{lossname}.loss"] = np.array(loss.item())
                pin = stats.pop("pool_in")
                if lossname == "infonce":   # the forward is the same for both losses
                    out[f"{tag}.{prec}.embedding"] = emb.detach().double().numpy()
                    out[f"{tag}.{prec}.pool_in_samples"] = pin.flatten()[cases.sample_idx(pin.numel(), 4096, 21)].double().numpy()
                    out[f"{tag}.pool_in_shape"] = np.array(pin.shape)
                    for k, (mean, var) in stats.items():
                        out[f"{tag}.{prec}.{k}.batch_mean"] = mean.double().numpy()
                        out[f"{tag}.{prec}.{k}.batch_var"] = var.double().numpy()
                    for n, b in m.named_buffers():
                        if "running" in n:
                            out[f"{tag}.{prec}.buf.{n}"] = b.double().numpy()
                names = []
                for j, (n, q) in enumerate(m.named_parameters()):
                    g = q.grad.detach().double().flatten()
                    names.append(n)
                    gs = g[cases.sample_idx(g.numel(), 512, 1000 + j)].numpy()
                    out[f"{tag}.{prec}.{lossname}.grad_samples.{n}"] = gs if prec == "f64" else gs.astype(np.float32)
                    out[f"{tag}.{prec}.{lossname}.grad_norm.{n}"] = np.array([float(g.norm()), float(g.abs().max()), float(g.sum())])
                out[f"{tag}.param_names"] = np.array(names)
                print(f"  train[{tag}, {prec}, {lossname}]: loss {loss.item():.6f}", flush=True)
    np.savez_compressed(os.path.join(HERE, "train.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["fbanks", "features", "logmel", "encoder", "infonce", "augment", "song_a", "dataset", "train"]
    for w in which:
        print("generating", w, flush=True)
        globals()["gen_" + w]()
    print("done")
