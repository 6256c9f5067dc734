"""GPU: the reference trainer's call contract, restated line for line, runs on this package with ONLY the three import
swaps of INTEGRATION.md section 2 (FMABaselineDataset/baseline_collate_fn, MixingStyleEncoder, InfoNCELoss).

What is restated (reference barry-mir/mixing-style-transfer, src/train.py):
  :460-469  FMABaselineDataset(...) with the reference's keyword arguments only
  :496-507  DataLoader(..., collate_fn=baseline_collate_fn, prefetch_factor=2, multiprocessing_context='fork')
  :522-524  feature_dim = full_dataset[0][1][0].shape[0]
  :545-555  MixingStyleEncoder(..., feature_dim=feature_dim).to(device)
  :223-262,:297-325  train_epoch body, fp32 branch: unpack, .to(device), the isnan checks, zero_grad, model(stems_dict,
            mixing_features), criterion, backward, optimizer.step
  :388-427  validate_epoch body
The stems are RIFF bytes stored under the `{stem}.mp3` names the reference hard-codes (no mp3 decoder offline).
"""
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

import cases

pytestmark = pytest.mark.gpu


def _count_stage_a(monkeypatch):
    from mst_amd.mixing_utils import MelFeatPlan
    calls = []
    real = MelFeatPlan.forward_stems

    def counted(self, stems_dict, want_logmel=True, want_feats=True, *a, **k):
        calls.append((want_logmel, want_feats, next(iter(stems_dict.values())).shape[0]))
        return real(self, stems_dict, want_logmel, want_feats, *a, **k)
    monkeypatch.setattr(MelFeatPlan, "forward_stems", counted)
    return calls


def test_reference_train_and_validate_loops_run_unchanged(tmp_path, monkeypatch):
    # ---- the three import swaps; everything below follows src/train.py
    from mst_amd.data import FMABaselineDataset, baseline_collate_fn
    from mst_amd.loss import InfoNCELoss
    from mst_amd.model import MixingStyleEncoder

    class args:   # src/params.py defaults, clip shortened to the toy tracks
        separated_path = cases.write_toy_tracks(str(tmp_path))
        clip_duration, sample_rate, n_fft, hop_length, n_mels = 0.25, 44100, 1024, 256, 128
        band_split_size, band_overlap, encoder_dim = 20, 10, 768
        batch_size, num_workers, learning_rate, weight_decay, temperature = 5, 2, 1e-4, 0.01, 0.1
        use_adversarial = False

    device = torch.device("cuda")
    torch.manual_seed(42)
    np.random.seed(42)
    full_dataset = FMABaselineDataset(                                           # train.py:460-469
        separated_path=args.separated_path, clip_duration=args.clip_duration, sample_rate=args.sample_rate,
        n_fft=args.n_fft, hop_length=args.hop_length, n_mels=args.n_mels, num_segments=2, min_audio_duration=25.0)
    train_dataloader = DataLoader(                                               # train.py:496-507
        full_dataset, batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers,
        collate_fn=baseline_collate_fn, pin_memory=False, prefetch_factor=2 if args.num_workers > 0 else None,
        persistent_workers=False, multiprocessing_context='fork' if args.num_workers > 0 else None)
    stems_list, features_list, song_idx, track_dir = full_dataset[0]            # train.py:522-524
    feature_dim = features_list[0].shape[0]
    assert feature_dim == 64
    model = MixingStyleEncoder(                                                  # train.py:545-555
        sample_rate=args.sample_rate, n_fft=args.n_fft, hop_length=args.hop_length, n_mels=args.n_mels,
        split_size=args.band_split_size, overlap=args.band_overlap, channels=8, embed_dim=args.encoder_dim,
        feature_dim=feature_dim).to(device)
    optimizer = torch.optim.AdamW(model.parameters(), lr=args.learning_rate, weight_decay=args.weight_decay)
    criterion = InfoNCELoss(temperature=args.temperature)
    calls = _count_stage_a(monkeypatch)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}

    # ---- train_epoch body (train.py:211-262, fp32 branch :297-325)
    model.train()
    losses, batches = [], 0
    for epoch in range(2):
        for batch_idx, batch_data in enumerate(train_dataloader):
            if len(batch_data) == 4:
                stems_dict, mixing_features, song_labels, track_dirs = batch_data
            else:
                stems_dict, mixing_features, song_labels = batch_data
                track_dirs = None
            stems_dict = {k: v.to(device) for k, v in stems_dict.items()}
            for k, v in stems_dict.items():
                assert not torch.isnan(v).any()
            assert not torch.isnan(mixing_features).any()          # train.py:237 (a None here was round 1's TypeError)
            assert not torch.isnan(song_labels.float()).any()
            mixing_features = mixing_features.to(device)
            song_labels = song_labels.to(device)
            assert mixing_features.shape == (10, 64) and stems_dict["vocals"].shape == (10, 2, 11025)
            optimizer.zero_grad()
            embeddings = model(stems_dict, mixing_features)
            loss_contrastive = criterion(embeddings, song_labels)
            loss = loss_contrastive
            loss.backward()
            optimizer.step()
            losses.append(loss.detach().item())
            batches += 1
    assert batches == 2 and all(np.isfinite(losses)) and embeddings.shape == (10, 768)
    assert calls == [(True, True, 10)] * 2, calls            # stage A ran ONCE per batch, features + log-mel together
    moved = [k for k, v in model.named_parameters() if not torch.equal(v.detach(), before[k])]
    assert len(moved) > 0.9 * len(before), f"only {len(moved)} of {len(before)} parameters changed"
    assert len(model._warned) == 0, model._warned                  # the hand-written trunk took the training calls

    # ---- validate_epoch body (train.py:388-427) + what the contract call must equal
    from mst_amd.mixing_utils import FEATURES_DEFERRED, MixingFeatureExtractor
    from oracle import encoder as oenc
    from oracle import features as ofeat
    val_dataloader = DataLoader(full_dataset, batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers,
                                collate_fn=baseline_collate_fn, pin_memory=False, prefetch_factor=2,
                                persistent_workers=False, multiprocessing_context='fork')
    model.eval()
    ext = MixingFeatureExtractor(args.sample_rate, args.n_fft, args.hop_length, args.n_mels)
    with torch.no_grad():
        for batch_idx, batch_data in enumerate(val_dataloader):
            stems_dict, mixing_features, song_labels, track_dirs = batch_data
            assert bool((mixing_features == FEATURES_DEFERRED).all())     # what fork'd workers hand out
            stems_dict = {k: v.to(device) for k, v in stems_dict.items()}
            mixing_features = mixing_features.to(device)
            song_labels = song_labels.to(device)
            embeddings = model(stems_dict, mixing_features)
            loss = criterion(embeddings, song_labels)
            assert np.isfinite(loss.item())
            # bit for bit what the explicit two-call form gives (same kernels, same launch shapes)
            feats, logmel = ext.features_and_logmel(stems_dict)
            assert torch.equal(embeddings, model.forward_from_logmel(logmel, feats))
            # real features passed in are used as given: rows 0..4 real (perturbed), rows 5..9 deferred
            mixed = mixing_features.clone()
            mixed[:5] = feats[:5] + 0.25
            e2 = model(stems_dict, mixed)
            want = model.forward_from_logmel(logmel, torch.cat([feats[:5] + 0.25, feats[5:]]))
            assert torch.equal(e2, want) and not torch.equal(e2[:5], embeddings[:5]) and torch.equal(e2[5:], embeddings[5:])
            # and the CPU oracle on the same clips
            x = torch.cat([stems_dict[s] for s in cases.STEMS], 1).cpu()
            sd = {k: v.cpu() for k, v in model.state_dict().items()}
            ref = oenc.encoder_forward(sd, x, ofeat.extract_all_features(x))
            err = (embeddings.cpu() - ref).abs().max().item() / ref.abs().max().item()
            print(f"contract call vs CPU oracle: max |d emb| / max |emb| = {err:.2e}")
            assert err < 2e-4


def test_deferred_rows_need_a_feature_layout():
    from mst_amd.mixing_utils import deferred_features, detailed_bins_for_feature_dim
    from mst_amd.model import MixingStyleEncoder
    assert detailed_bins_for_feature_dim(64) == 0 and detailed_bins_for_feature_dim(180) == 32
    assert detailed_bins_for_feature_dim(63) is None and detailed_bins_for_feature_dim(52) is None
    m = MixingStyleEncoder(feature_dim=63).cuda().eval()
    x = cases.synth_clip(0, 11025)[None].cuda()
    stems = {s: x[:, 2 * i:2 * i + 2] for i, s in enumerate(cases.STEMS)}
    with torch.no_grad():
        assert m(stems, torch.zeros(1, 63, device="cuda")).shape == (1, 768)     # real features: fine
        with pytest.raises(ValueError):
            m(stems, deferred_features(63)[None].cuda())
    # detailed-spectral layout (feature_dim 180): the model's own stage-A launch emits that layout
    from mst_amd.mixing_utils import MixingFeatureExtractor
    m = MixingStyleEncoder(feature_dim=180).cuda().eval()
    ext = MixingFeatureExtractor(use_detailed_spectral=True, n_spectral_bins=32)
    with torch.no_grad():
        feats, logmel = ext.features_and_logmel(stems)
        assert feats.shape == (1, 180)
        assert torch.equal(m(stems, deferred_features(180)[None].cuda()), m.forward_from_logmel(logmel, feats))


def test_reference_amp_branch_runs_unchanged(tmp_path):
    """src/train.py:246-296, the `--use_amp` branch (scaler is not None): forward and loss under torch.cuda.amp.autocast,
    scaler.scale(loss).backward(), scaler.step, scaler.update -- restated on this package.  Inside autocast(float16) the
    hand-written conv trunk switches to its float16-operand kernels by itself (`train_precision = "auto"`: forward, input
    gradient and weight gradients on v_mfma_f32_16x16x32_f16 with fp32 accumulation, see include/mst.h); the FiLM MLP and the
    attention head run under autocast as torch modules; the InfoNCE kernels take the half-precision embeddings.  Clips of
    0.25 s (44 frames) and a batch of 10: ragged 8-clip groups and single-tile planes in every kernel."""
    from mst_amd.data import FMABaselineDataset, baseline_collate_fn
    from mst_amd.loss import InfoNCELoss
    from mst_amd.model import MixingStyleEncoder
    device = torch.device("cuda")
    torch.manual_seed(0)
    np.random.seed(0)
    ds = FMABaselineDataset(separated_path=cases.write_toy_tracks(str(tmp_path)), clip_duration=0.25, sample_rate=44100,
                            n_fft=1024, hop_length=256, n_mels=128, num_segments=2, min_audio_duration=25.0)
    dl = DataLoader(ds, batch_size=5, shuffle=True, num_workers=2, collate_fn=baseline_collate_fn, pin_memory=False,
                    prefetch_factor=2, persistent_workers=False, multiprocessing_context='fork')
    model = MixingStyleEncoder(sample_rate=44100, n_fft=1024, hop_length=256, n_mels=128, split_size=20, overlap=10,
                               channels=8, embed_dim=768, feature_dim=ds[0][1][0].shape[0]).to(device)
    optimizer = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01)
    criterion = InfoNCELoss(temperature=0.1)
    scaler = torch.cuda.amp.GradScaler()
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    model.train()
    for batch_data in dl:
        stems_dict, mixing_features, song_labels, track_dirs = batch_data
        stems_dict = {k: v.to(device) for k, v in stems_dict.items()}
        mixing_features = mixing_features.to(device)
        song_labels = song_labels.to(device)
        optimizer.zero_grad()
        with torch.cuda.amp.autocast():
            embeddings = model(stems_dict, mixing_features)
            loss_contrastive = criterion(embeddings, song_labels)
            loss = loss_contrastive
        scaler.scale(loss).backward()
        scaler.step(optimizer)
        scaler.update()
        assert np.isfinite(loss.detach().item()) and embeddings.shape == (10, 768)
    moved = [k for k, v in model.named_parameters() if not torch.equal(v.detach(), before[k])]
    assert len(moved) > 0.9 * len(before) and len(model._warned) == 0
    assert model._hip_train.train_f16   # the f16 kernels did run
    assert all(torch.isfinite(v).all() for v in model.parameters())
