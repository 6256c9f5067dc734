"""CPU: Dataset / collate host logic vs the oracle restatement, and the N>1 exchange path (world_size 2, gloo)."""
import os
import wave

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from oracle import dataset as odata
from oracle import loss as oloss


def write_wav(path, x, sr=44100):
    a = (x.clamp(-1, 1).T.numpy() * 32767.0).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(x.shape[0]); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes(a.tobytes())


@pytest.fixture(scope="module")
def stem_dir(tmp_path_factory):
    root = tmp_path_factory.mktemp("sep")
    lens = {"long": 9000, "short": 5000, "mono": 9000}
    for name, L in lens.items():
        d = root / name
        d.mkdir()
        x = cases.synth_clip(len(name), L, 4000)
        for i, s in enumerate(cases.STEMS):
            write_wav(str(d / f"{s}.wav"), x[2 * i:2 * i + (1 if name == "mono" else 2)], 4000)
    return str(root)


def make_ds(stem_dir, **kw):
    from mst_amd.data import FMABaselineDataset
    return FMABaselineDataset(stem_dir, clip_duration=1.0, sample_rate=4000, stem_ext=".wav", compute_features=False, **kw)


def test_dataset_matches_oracle_crops_and_collate(stem_dir):
    from mst_amd.data import baseline_collate_fn
    ds = make_ds(stem_dir)
    assert len(ds) == 3 and ds.clip_samples == 4000
    np.random.seed(42)
    items = [ds[i] for i in range(len(ds))]
    np.random.seed(42)
    for (stems_list, feats_list, idx, tdir) in items:
        L = 5000 if tdir.endswith("short") else 9000
        starts = odata.crop_starts(L, 4000, 2)
        assert len(stems_list) == 2 and feats_list == [None, None]
        full = ds._load_stems(tdir)
        for clip, st in zip(stems_list, starts):
            ref = odata.extract_clip(full, st, 4000)
            for s in cases.STEMS:
                assert clip[s].shape == (2, 4000) and torch.equal(clip[s], ref[s])
        if tdir.endswith("short"):          # audio shorter than 2 clips: both start at 0 (src/data.py:241-248)
            assert starts == [0, 0] and torch.equal(stems_list[0]["bass"], stems_list[1]["bass"])
        if tdir.endswith("mono"):           # mono stems are duplicated to stereo (src/data.py:178-180)
            assert torch.equal(stems_list[0]["drums"][0], stems_list[0]["drums"][1])
    sd, feats, labels, dirs = baseline_collate_fn(items)
    osd, _, olabels, odirs = odata.collate([(a, [torch.zeros(1)] * 2, c, d) for a, b, c, d in items])
    assert feats is None and torch.equal(labels, olabels) and dirs == odirs and labels.dtype == torch.int64
    for s in cases.STEMS:
        assert sd[s].shape == (6, 2, 4000) and torch.equal(sd[s], osd[s])


def test_dataset_single_segment_padding_and_errors(stem_dir, tmp_path):
    from mst_amd.data import FMABaselineDataset, SCNetSeparator
    ds = make_ds(stem_dir, num_segments=1)
    ds.clip_samples = 6000                      # longer than the 'short' track -> zero padded tail (src/data.py:283-287)
    i = [k for k, d in enumerate(ds.track_dirs) if d.endswith("short")][0]
    stems_list, _, _, _ = ds[i]
    assert stems_list[0]["other"].shape == (2, 6000) and not stems_list[0]["other"][:, 5000:].any()
    with pytest.raises(ValueError):
        FMABaselineDataset(str(tmp_path / "nope"))
    with pytest.raises(ValueError):
        make_ds(stem_dir, num_segments=3)[0]
    os.makedirs(tmp_path / "sep2" / "t0")
    with pytest.raises(FileNotFoundError):
        FMABaselineDataset(str(tmp_path / "sep2"), stem_ext=".wav")[0]
    with pytest.raises(RuntimeError):
        SCNetSeparator("model.ckpt", "config.yaml")


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mst_amd.data import shard_batch
        from mst_amd.loss import InfoNCELoss, gather_embeddings
        g = torch.Generator().manual_seed(11)
        emb = torch.randn(12, 32, generator=g)
        labels = torch.arange(12) // 2          # pairs: both segments of a song are adjacent
        stems = {s: torch.arange(12.0)[:, None, None].repeat(1, 2, 8) for s in cases.STEMS}
        sd, _, lab = shard_batch(stems, None, labels, rank, world)
        lo = rank * 12 // world
        assert sd["bass"].shape[0] == 12 // world and sd["bass"][0, 0, 0].item() == lo
        local = emb[lo:lo + 12 // world].clone().requires_grad_(True)
        all_e, all_l, row0 = gather_embeddings(local, lab)
        assert row0 == lo and torch.equal(all_l, labels) and torch.allclose(all_e.detach(), emb)
        loss = InfoNCELoss(0.1, gather=True)(local, lab)
        loss.backward()
        # reference on the un-sharded batch
        full = emb.clone().requires_grad_(True)
        ref = oloss.info_nce(full, labels, 0.1)
        ref.backward()
        # every rank returns the global loss; gathered local gradients == gradient of the un-sharded reference loss
        gl = [torch.zeros(12 // world, 32) for _ in range(world)]
        dist.all_gather(gl, local.grad)
        ret[rank] = (abs(loss.item() - ref.item()) < 1e-5 and torch.allclose(torch.cat(gl), full.grad, atol=1e-6))
        # forward-only (no grad): identical scalar on every rank
        v = InfoNCELoss(0.1, gather=True)(local.detach(), lab)
        ret[rank] = ret[rank] and abs(v.item() - ref.item()) < 1e-5
    finally:
        dist.destroy_process_group()


def test_sharded_gather_infonce_world2_gloo():
    world, port = 2, 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


# ---------------------------------------------------------------------------------------------------------------
# PCM shard ingest
# ---------------------------------------------------------------------------------------------------------------
def _toy_track(L, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(8, L, generator=g) * 2 - 1) * 0.9


def test_pcm_shard_roundtrip_and_errors(tmp_path):
    from mst_amd import ingest
    x = _toy_track(5000, 1)
    p = str(tmp_path / "a.pcm16")
    ingest.write_pcm_shard(p, x, 44100)
    mm, sr = ingest.open_pcm_shard(p)
    assert sr == 44100 and mm.shape == (8, 5000)
    q = ingest.float_to_pcm16(x)
    assert np.array_equal(np.asarray(mm), q.numpy())
    assert (q.float() / 32768.0 - x).abs().max().item() <= 0.5 / 32768 + 1e-7
    # saturation at full scale, dict input in reference stem order
    d = {s: torch.full((2, 10), v) for s, v in zip(("vocals", "bass", "drums", "other"), (1.0, -1.0, 0.25, 0.0))}
    ingest.write_pcm_shard(p, d)
    mm, _ = ingest.open_pcm_shard(p)
    assert mm[0, 0] == 32767 and mm[2, 0] == -32768 and mm[4, 0] == 8192 and mm[6, 0] == 0
    with open(p, "r+b") as f:
        f.write(b"XXXX")
    with pytest.raises(ValueError):
        ingest.open_pcm_shard(p)
    with pytest.raises(ValueError):
        ingest.PcmShardDataset(str(tmp_path / "missing"))


def test_pcm_shard_dataset_matches_reference_sampling(tmp_path):
    """Same numpy-RNG crop starts, zero padding and collate order as FMABaselineDataset (src/data.py:201-328)."""
    from mst_amd import ingest
    from oracle import dataset as odata
    sr, dur = 1000, 1.0
    lengths = [3500, 1500, 2000, 700]           # >2C, <2C, ==2C, <C
    tracks = [_toy_track(L, 10 + i) for i, L in enumerate(lengths)]
    for i, t in enumerate(tracks):
        ingest.write_pcm_shard(str(tmp_path / f"t{i}.pcm16"), t, sr)
    for nseg in (1, 2):
        ds = ingest.PcmShardDataset(str(tmp_path), clip_duration=dur, sample_rate=sr, num_segments=nseg)
        assert len(ds) == 4
        np.random.seed(42)
        items = [ds[i] for i in range(4)]
        np.random.seed(42)
        want = [odata.crop_starts(L, int(dur * sr), nseg) for L in lengths]
        for (clips, idx, path), starts, t in zip(items, want, tracks):
            assert len(clips) == nseg
            q = ingest.float_to_pcm16(t)
            for c, s in zip(clips, starts):
                ref = torch.zeros(8, 1000, dtype=torch.int16)
                n = max(0, min(1000, q.shape[1] - s))
                ref[:, :n] = q[:, s:s + n]
                assert torch.equal(c, ref)
        stems, labels, paths = ingest.pcm_collate_fn(items)
        assert stems.shape == (4 * nseg, 8, 1000) and stems.dtype == torch.int16
        assert labels.tolist() == [i for i in range(4) for _ in range(nseg)]
        v = ingest.stems_views(stems)
        assert v["drums"].shape == (4 * nseg, 2, 1000) and v["drums"].data_ptr() == stems[:, 4:6].data_ptr()
    with pytest.raises(ValueError):
        ingest.PcmShardDataset(str(tmp_path), num_segments=3)[0]


# ---------------------------------------------------------------------------------------------------------------
# Retrieval validation + style-transfer sampling
# ---------------------------------------------------------------------------------------------------------------
def test_retrieval_metrics_match_reference_loop():
    from mst_amd import validation_utils as vu
    from oracle import retrieval as oret
    g = torch.Generator().manual_seed(3)
    pool = torch.randn(40, 32, generator=g)
    pool_idx = list(range(100, 140))
    q_idx = [100 + i for i in (0, 5, 7, 11, 39, 20, 21)]
    queries = torch.stack([pool[i - 100] + 0.9 * torch.randn(32, generator=g) for i in q_idx])
    want = oret.evaluate_retrieval_accuracy(queries, pool, q_idx, pool_idx, (1, 3, 5))
    got = vu.evaluate_retrieval_accuracy(queries, pool, q_idx, pool_idx, [1, 3, 5])
    assert got == want and 0.0 < got["top_1_accuracy"] <= got["top_5_accuracy"] <= 1.0
    i, s = vu.retrieve_top_k(queries[0], pool, k=5)
    oi, os_ = oret.retrieve_top_k(queries[0], pool, k=5)
    assert torch.equal(i, oi) and torch.allclose(s, os_) and (s[:-1] >= s[1:]).all()


def test_style_transfer_dataset_sampling_and_collate(stem_dir):
    from mst_amd.data import StyleTransferDataset, style_transfer_collate_fn
    from oracle import retrieval as oret
    ds = StyleTransferDataset(None, stem_dir, clip_duration=1.5, sample_rate=4000, stem_ext=".wav", compute_features=False)
    assert len(ds) == 3 and ds.clip_samples == 6000
    lengths = [5000 if d.endswith("short") else 9000 for d in ds.track_dirs]
    np.random.seed(7)
    items = [ds[i] for i in (0, 1, 2, 1)]
    np.random.seed(7)
    for (inp, tgt, feats), idx in zip(items, (0, 1, 2, 1)):
        a, t, b = oret.style_transfer_draws(idx, lengths, 6000)
        assert feats is None and t != idx
        for stems, track, start in ((inp, idx, a), (tgt, t, b)):
            full = ds._load_full(track)
            for s in cases.STEMS:
                assert stems[s].shape == (2, 6000)
                if start is None:       # short track: zero padded, no RNG draw
                    assert torch.equal(stems[s][:, :5000], full[s]) and not stems[s][:, 5000:].any()
                else:
                    assert torch.equal(stems[s], full[s][:, start:start + 6000])
    i, t, f = style_transfer_collate_fn(items)
    assert f is None and i["vocals"].shape == (4, 2, 6000) and torch.equal(t["other"][3], items[3][1]["other"])
    with pytest.raises(ValueError):
        StyleTransferDataset(None, stem_dir + "_missing")
    with pytest.raises(ValueError):
        StyleTransferDataset("x", stem_dir, use_preseparated=False)


def test_load_stems_segment_dirs_and_shards(stem_dir, tmp_path):
    from mst_amd import ingest, validation_utils as vu
    from mst_amd.data import FMABaselineDataset
    ds = make_ds(stem_dir)
    d = [t for t in ds.track_dirs if t.endswith("short")][0]
    seg = vu.load_stems_segment(d, 1.0, 0.5, 4000, stem_ext=".wav")
    full = ds._load_stems(d)
    for s in cases.STEMS:
        assert seg[s].shape == (2, 2000) and np.array_equal(seg[s], full[s][:, 4000:6000].numpy()[:, :2000] if full[s].shape[1] >= 6000
                                                           else np.pad(full[s][:, 4000:].numpy(), ((0, 0), (0, 1000))))
    p = str(tmp_path / "short.pcm16")
    ingest.write_pcm_shard(p, full, 4000)
    seg2 = vu.load_stems_segment(p, 1.0, 0.5, 4000)
    for s in cases.STEMS:   # wav fixtures are 16-bit: the shard reproduces them to within one LSB of the 32767/32768 scale
        assert np.abs(seg2[s] - seg[s]).max() <= 1.5 / 32768
    with pytest.raises(FileNotFoundError):
        vu.load_stems_segment(str(tmp_path), 0.0, 1.0, 4000)
