"""Minimal contrastive training loop on the MI355X path -- what the reference's src/train.py:223-332 does per step,
with this package's drop-ins:

    PCM shards -> PcmShardDataset / pcm_collate_fn (fork'd DataLoader workers, int16, pinned)
               -> DeviceStager (async H2D, overlapped)
               -> stage A in HIP: 64-d mixing features + log-mel straight from the int16 batch
               -> MixingStyleEncoder (train mode: PyTorch-ROCm autograd for the encoder; eval mode: all HIP)
               -> InfoNCELoss (forward and backward in HIP)
               -> AdamW

    python examples/train_contrastive.py --shards /data/shards --steps 100 --batch-size 24

With --synthetic N the script first writes N synthetic tracks as shards into --shards (no dataset needed).
"""
import argparse
import os
import sys

import numpy as np
import torch
from torch.utils.data import DataLoader

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mst_amd import ingest  # noqa: E402
from mst_amd.loss import InfoNCELoss  # noqa: E402
from mst_amd.mixing_utils import deferred_features  # noqa: E402
from mst_amd.model import MixingStyleEncoder  # noqa: E402
from mst_amd.synth import synth_clip  # noqa: E402


def write_synthetic_shards(path, n_tracks, seconds, sr):
    os.makedirs(path, exist_ok=True)
    for i in range(n_tracks):
        ingest.write_pcm_shard(os.path.join(path, f"track{i:04d}.pcm16"), synth_clip(1000 + i, int(seconds * sr), sr), sr)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", required=True)
    ap.add_argument("--synthetic", type=int, default=0, help="write this many synthetic tracks into --shards first")
    ap.add_argument("--track-seconds", type=float, default=25.0)
    ap.add_argument("--clip-seconds", type=float, default=10.0)
    ap.add_argument("--batch-size", type=int, default=24, help="songs per batch (2 segments each)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--train-backend", choices=["hip", "torch"], default="hip",
                    help="hip: conv trunk forward + backward in libmst.so (default); torch: PyTorch-ROCm autograd")
    ap.add_argument("--train-precision", choices=["fp32", "f16x3", "f16"], default="fp32",
                    help="precision of the hand-written conv trunk: fp32 = exact fp32 MFMA; f16x3 = 3-term split-precision f16, "
                         "fp32-equivalent gradients at about twice the speed; f16 = float16 operands with fp32 accumulation (the "
                         "arithmetic of the reference's --use_amp step)")
    ap.add_argument("--small-nets", choices=["hip", "torch"], default="hip",
                    help="pooling head and FiLM MLP of the training step: hand-written forward / backward (csrc/head.hip) or the nn.Modules")
    a = ap.parse_args(argv)
    sr = 44100
    if a.synthetic:
        write_synthetic_shards(a.shards, a.synthetic, a.track_seconds, sr)
    torch.manual_seed(a.seed)
    np.random.seed(a.seed)
    dev = torch.device("cuda")
    ds = ingest.PcmShardDataset(a.shards, clip_duration=a.clip_seconds, sample_rate=sr, num_segments=2)
    dl = DataLoader(ds, batch_size=a.batch_size, shuffle=True, drop_last=True, num_workers=a.workers,
                    collate_fn=ingest.pcm_collate_fn, pin_memory=True)
    model = MixingStyleEncoder(sr, 1024, 256, 128, 20, 10, 8, 768, feature_dim=64).to(dev).train()
    model.train_backend = a.train_backend
    model.train_precision = a.train_precision
    model.small_nets_backend = a.small_nets
    deferred = torch.stack([deferred_features(64)] * (2 * a.batch_size)).to(dev)   # what the Dataset's feature slot carries
    crit = InfoNCELoss(0.1)   # (check="deferred" would drop the per-step read-back of the guard; this loop reads loss.item() anyway)
    opt = torch.optim.AdamW(model.parameters(), lr=a.lr)
    stager = ingest.DeviceStager((2 * a.batch_size, 8, ds.clip_samples), torch.int16, dev)
    losses, step = [], 0
    it = iter(dl)
    stems, labels, _ = next(it)
    fut, lab = stager.submit(stems), labels
    while step < a.steps:
        x = fut.get()
        cur_lab = lab.to(dev, non_blocking=True)
        try:                                   # stage the next batch while this one is on the GPU
            stems, labels, _ = next(it)
        except StopIteration:
            it = iter(dl)
            stems, labels, _ = next(it)
        nxt, lab = stager.submit(stems), labels
        # the reference trainer's call (src/train.py:253) with the Dataset's deferred feature rows: ONE stage-A launch inside
        # yields the features and the log-mel, in the layout the training trunk reads
        emb = model(ingest.stems_views(x), deferred[:x.shape[0]])
        stager.release(fut)
        loss = crit(emb, cur_lab)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        step += 1
        fut = nxt
        if step % 10 == 0 or step == a.steps:
            print(f"step {step}: loss {losses[-1]:.4f}", flush=True)
    return losses


if __name__ == "__main__":
    main()
