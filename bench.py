#!/usr/bin/env python3
"""Contract benchmark: triplets/s of the contrastive data-path hot path on N MI355X (one process per GPU).

A "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
    24 triplets = 72 clips of 10 s / 44.1 kHz / 4 stems x stereo per GPU  (BASELINE.json configs[1]/[2])
    stage A  HIP  waveform -> STFT -> 128-mel -> log-mel (B,8,128,1723) + 64-d mixing features
    stage B       FiLM MLP + 11 x band-split Conv/BN/FiLM/ReLU/MaxPool x2 + attention pooling -> (B,768)
                  ("hip": libmst.so fp32-MFMA kernels = configs[2];  "torch": PyTorch-ROCm ops = configs[1])
    exchange      all-gather of embeddings + labels over RCCL (N>1) and InfoNCE on the gathered batch
Prints ONE JSON line on rank 0 (see the driver contract in the task statement), with `roofline` for the
dominant hand-written kernel (HIP-event timed inside the timed region) and `cpu_baseline` (the CPU oracle
timed on the host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
MFMA_F32_PEAK_TF = 157.3     # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)
MFMA_F16_PEAK_TF = 2500.0    # dense f16 / bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense; the 5 PF headline is 2:1 sparsity)
MFMA_F16_SUSTAINED_TF = 2000.0   # what a pure f16 MFMA stream holds on operands that toggle (scripts/ubench_mfma.hip mode 4,
                                 # profiles/r04_ubench_mfma_corun.txt: the chip clocks down to ~1.9 GHz); reported beside the peak, never instead of it


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--triplets", type=int, default=24, help="triplets per GPU per step (3 clips each)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--encoder", choices=["hip", "torch"], default="hip",
                    help="hip (default, configs[2]): stage B on libmst.so's hand-written kernels; a missing / refusing HIP encoder is "
                         "an error, never a library fallback.  torch: BASELINE configs[1], stage B on PyTorch-ROCm (explicit opt-in)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the post-timing correctness check of the timed workload (config.verified)")
    ap.add_argument("--verify-b1", action="store_true",
                    help="config.verified: besides the re-ordered batches, also run every clip alone (B = 1) and compare bit for bit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["fp32", "f16x3", "f16x3-all", "f16"], default="fp32",
                    help="conv1 arithmetic: exact fp32 MFMA (default) or opt-in 3-term split-precision f16 MFMA")
    ap.add_argument("--aug", action="store_true", help="also run the augmentation chain on the negative clip of every triplet inside the timed step (BASELINE configs[3] without SCNet)")
    ap.add_argument("--config", choices=["default", "baseline_sh", "config5"], default="default",
                    help="default: BASELINE configs[1]/[2] (1024/256/128 mels, 20/10 sub-bands, 768-d). Other shapes are "
                         "extra measurements, not the contract line: baseline_sh = the reference's scripts/train_baseline.sh "
                         "(2048/512/80 mels, 16/8, 512-d); config5 = BASELINE configs[4] shapes (256 mels, 24 sub-bands; "
                         "use --seconds 30 --triplets 8), forward only, fp32")
    ap.add_argument("--train", action="store_true",
                    help="extra measurement, not the contract line: a full TRAINING step (stage A, encoder forward + backward "
                         "with train-mode BatchNorm and Dropout, InfoNCE forward + backward, AdamW) instead of the forward path")
    ap.add_argument("--train-backend", choices=["hip", "torch"], default="hip")
    ap.add_argument("--sync-bn", action="store_true",
                    help="--train with --gpus N > 1: the ranks add up their BatchNorm statistics (exact integer sums over RCCL), so "
                         "that the N-GPU step is the single-process step on the whole batch; default: per-rank statistics")
    ap.add_argument("--train-precision", choices=["fp32", "f16", "f16x3", "amp"], default="fp32",
                    help="--train only.  fp32: exact fp32 MFMA trunk.  f16: the hand-written trunk with float16 operands / fp32 "
                         "accumulation (BASELINE configs[4]'s fp16), everything else fp32.  f16x3: the same kernels with 3-term split-precision "
                         "operands (fp32-equivalent results).  amp: the reference's --use_amp step "
                         "(src/train.py:246-262): forward and loss under torch.autocast(float16), GradScaler; the trunk "
                         "switches to its f16 kernels by itself, the torch backend runs MIOpen's half convolutions")
    ap.add_argument("--logmel-layout", choices=["auto", "ref"], default="auto",
                    help="auto (default): stage A hands the log-mel to the HIP encoder in its internal channel-minor layout "
                         "(include/mst.h MST_LOGMEL_CM32; CM16 = float16 hi/lo planes for the f16 / split-precision modes) -- same "
                         "values, same bytes, whole-line stores; ref: the reference's (B, 8, n_mels, frames) tensor (A/B)")
    ap.add_argument("--ingest", choices=["resident", "f32", "pcm16"], default="resident",
                    help="resident (default, the contract: inputs in HBM before timing) | f32 | pcm16: every step's batch "
                         "comes from pinned host memory over PCIe (double-buffered, overlapped); PCIe-inclusive rate")
    return ap.parse_args()


def cpu_baseline(model_sd, x, cfg, budget_s=15.0):
    """CPU oracle (the parity-pinned restatement of the reference path) on the host cores, bounded sample.
    x: (3, 8, T) fp32 on the CPU -- three clips of the TIMED batch (first, middle, last), so that the embeddings this leg
    computes anyway double as the checker of the timed workload (returned second; see `verify`)."""
    from oracle import encoder as oenc
    from oracle import features as ofeat
    # the GPU box gives a 1-GPU job a share of 16 host cores; os.cpu_count() reports the whole machine
    ncpu = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(ncpu)
    sd = {k: v.detach().cpu() for k, v in model_sd.items()}
    B, T = x.shape[0], x.shape[-1]
    geo = (cfg["sr"], cfg["n_fft"], cfg["hop"], cfg["n_mels"])
    clips, t0 = 0, time.perf_counter()
    with torch.no_grad():
        oenc.encoder_forward(sd, x[:1], ofeat.extract_all_features(x[:1], *geo), *geo, cfg["split"], cfg["overlap"])  # warm-up (thread pools, fft plans)
        t0 = time.perf_counter()
        while True:
            f = ofeat.extract_all_features(x, *geo)                                  # reference: computed in the Dataset worker
            emb = oenc.encoder_forward(sd, x, f, *geo, cfg["split"], cfg["overlap"])  # reference: MixingStyleEncoder.forward (mel again + CNN)
            clips += B
            el = time.perf_counter() - t0
            if el > budget_s or clips >= 150:
                break
    return {"value": round(clips / 3.0 / el, 4), "unit": "triplets/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{clips} clips ({clips // 3} triplets) of {T / cfg['sr']:.0f} s (clips first / middle / last of "
            f"the timed batch), oracle/ features+mel+encoder fwd, torch-CPU fp32, {el:.1f} s wall"}, emb


def verify(model, fe, lay_fn, backend, last, b1=False):
    """Post-timing correctness check of the workload that was just TIMED, at its own size (the launch shapes of the 72 x 10 s
    batch take index paths no small parity case does).  Batch independence: the last timed step's batch is run again with its
    clips in REVERSED order and ROTATED by 29 -- launches of the timed shape, in which every clip sits at another position, in
    another tile set, next to other neighbours -- and every clip must reproduce its embedding BIT FOR BIT (HIP encoder; the
    library backend of configs[1] is held to 1e-4 of the embedding's maximum instead).  `b1` (--verify-b1; always on in
    tests/test_contract_size_gpu.py): additionally every clip alone (B = 1).  The default leaves the B = 1 launches out so
    that a `rocprofv3 --stats` run of this very command averages 72-clip launches only.  Returns the dict that becomes
    config.verified; `ok` False makes the bench exit non-zero.  `verify_oracle` adds the comparison with the CPU oracle."""
    from mst_amd import _lib as mlib
    emb, stems = last["emb"], last["stems"]
    B = emb.shape[0]
    out = {"clips": 0, "max_rel": None, "ok": True}
    if stems is None:   # --ingest: the staged batch is gone by now
        return out

    def forward(sd):
        lay = lay_fn()
        f1, lm1 = fe.features_and_logmel(sd, lay, lay == mlib.LOGMEL_CM16)
        return model.hip_encoder().forward(lm1, f1) if backend == "hip" else model.forward_from_logmel(lm1, f1)

    bad, worst, runs = set(), 0.0, []
    dev = emb.device
    perms = [("reversed", torch.arange(B - 1, -1, -1, device=dev)), ("rotated by 29", (torch.arange(B, device=dev) + 29) % B)] if B > 1 else []
    with torch.no_grad():
        for name, perm in perms:
            e = forward({k: v[perm].contiguous() for k, v in stems.items()})
            neq = (e != emb[perm]).any(dim=1)
            bad |= set(perm[neq].tolist())
            if bool(neq.any()):
                worst = max(worst, float(((e - emb[perm]).abs().amax(dim=1) / emb[perm].abs().amax(dim=1).clamp(min=1e-30)).max()))
            runs.append(name)
        if b1:
            for c in range(B):
                e1 = forward({k: v[c:c + 1] for k, v in stems.items()})
                if not torch.equal(e1[0], emb[c]):
                    bad.add(c)
                    worst = max(worst, float((e1[0] - emb[c]).abs().max() / emb[c].abs().max().clamp(min=1e-30)))
            runs.append("every clip alone (B = 1)")
    out["batch_independence"] = {"clips": B, "bit_equal": B - len(bad), "worst_rel_to_max": worst, "re_runs": runs}
    out["clips"] = B
    out["ok"] = len(bad) == 0 if backend == "hip" else worst <= 1e-4
    return out


def verify_oracle(out, emb, oracle_clips, oracle_emb, precision):
    """The embeddings of `oracle_clips` against the CPU oracle's (computed by the cpu_baseline leg on the same clips of the timed
    batch): every element within 1e-4 of max(|ref|, 1e-2 max|ref|) -- the element-wise bar of tests/test_encoder_gpu.py (plain f16
    operands, an opt-in mode outside the parity bar: 1e-3)."""
    tol = 1e-3 if precision == "f16" else 1e-4
    got = emb[oracle_clips].detach().cpu().double()
    ref = oracle_emb.double()
    rel = (got - ref).abs() / torch.maximum(ref.abs(), 1e-2 * ref.abs().amax(dim=1, keepdim=True))
    out["oracle"] = {"clips": list(oracle_clips), "max_rel": float(rel.max()), "tol": tol,
                     "beyond_tol": int((rel > tol).sum()), "elements": rel.numel(),
                     "normwise": float(((got - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max())}
    out["max_rel"] = out["oracle"]["max_rel"]
    out["ok"] = out["ok"] and out["oracle"]["beyond_tol"] == 0
    return out


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) as CHILD processes through
    torch.distributed.run and pass their output / exit code through.  This parent never touches the GPU
    (`torch.cuda.device_count()` does not initialise HIP on this image) and nothing is exec'd over it."""
    import socket
    import subprocess
    dry = bool(os.environ.get("MST_BENCH_DRYRUN"))
    have = torch.cuda.device_count()
    if not dry and not os.environ.get("MST_BENCH_ONE_GPU") and have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def dry_run(a, world, rank):
    """MST_BENCH_DRYRUN=1 (CPU rehearsal of the N-rank control flow, used by tests/test_bench_launcher_cpu.py): gloo
    rendezvous, rank count check, barrier + max-over-ranks timing of EMPTY steps, one JSON line on rank 0.  No kernels
    run and the line says so; it is never a measurement."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    seen = torch.ones(1)
    dist.all_reduce(seen)
    if dist.get_world_size() != a.gpus or int(seen.item()) != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but {dist.get_world_size()} ranks joined", file=sys.stderr)
        sys.exit(3)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pass
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN -- launcher rehearsal on gloo, no kernels, not a measurement", "value": 0.0,
                          "unit": "triplets/s", "n_gpus": world, "world": dist.get_world_size(),
                          "rccl_ranks_seen": int(seen.item()), "steps": a.steps, "warmup": a.warmup, "dry_run": True}),
              flush=True)
    dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:   # before ANY GPU call: spawn the N ranks, relay their result
        sys.exit(launch_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(3)
    if os.environ.get("MST_BENCH_DRYRUN"):
        return dry_run(a, world, rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    if not os.environ.get("MST_BENCH_ONE_GPU") and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        sys.exit(2)
    # host-side torch ops here are tiny (RNG draws, 22050-tap impulse responses): a 128-thread intra-op pool costs
    # milliseconds per op, so cap it at the box's per-GPU CPU share
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)) // max(1, world))))
    if os.environ.get("MST_BENCH_ONE_GPU"):   # rehearsal of the N>1 control flow on a 1-GPU box (all ranks on cuda:0)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MST_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only for the rehearsal
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)                       # every rank really takes part in a collective
        ranks_seen = int(seen.item())
        if dist.get_world_size() != a.gpus or ranks_seen != a.gpus:
            print(f"bench.py: --gpus {a.gpus} but {dist.get_world_size()} ranks joined ({ranks_seen} seen)", file=sys.stderr)
            sys.exit(3)
    else:
        ranks_seen = 1

    from mst_amd.loss import InfoNCELoss
    from mst_amd.mixing_utils import AudioAugmenter, MixingFeatureExtractor
    from mst_amd.model import MixingStyleEncoder
    from mst_amd.synth import synth_batch

    sr, n_fft, hop, n_mels = 44100, 1024, 256, 128
    split, overlap, embed = 20, 10, 768
    if a.config == "baseline_sh":
        n_fft, hop, n_mels, split, overlap, embed = 2048, 512, 80, 16, 8, 512
    elif a.config == "config5":
        n_mels = 256
    T = int(a.seconds * sr)
    B = 3 * a.triplets
    backend = a.encoder   # "hip" unless the caller asks for configs[1]: a HIP encoder that cannot run raises (rc != 0), nothing falls back

    torch.manual_seed(42)
    model = MixingStyleEncoder(sr, n_fft, hop, n_mels, split, overlap, 8, embed, feature_dim=64, encoder_backend=backend)
    with torch.no_grad():  # FiLM gammas ~ 1 (trained-looking) so activations stay O(1); random-init otherwise
        b = model.film_encoder.film_head.bias
        for i in range(model.audio_encoder.n_subbands):
            b[i * 192:i * 192 + 32] += 1.0
            b[i * 192 + 64:i * 192 + 128] += 1.0
    model = model.to(dev).eval()
    model.conv1_precision = a.precision
    fe = MixingFeatureExtractor(sr, n_fft, hop, n_mels)
    # the reference's "no positive pairs" guard is evaluated one call late from a pinned word (check="deferred"): the same
    # RuntimeError, without a device -> host read that would leave the GPU idle between steps (mst_amd/loss.py)
    crit = InfoNCELoss(0.1, gather=world > 1, check="deferred")
    augm = AudioAugmenter(sr, 9.0, 0.5)
    last = {}   # the last step's inputs and embeddings (for the post-timing check)
    torch.manual_seed(1234 + rank)

    x = synth_batch(B, T, sr, device=dev, first_clip=rank * B)   # resident in HBM before timing
    stems = {s: x[:, 2 * i:2 * i + 2] for i, s in enumerate(("vocals", "bass", "drums", "other"))}
    labels = (torch.arange(B, device=dev) // 3) + rank * a.triplets  # 3 clips of a triplet share a song id
    xa = x.clone() if a.aug else None   # batch whose negatives are overwritten by their augmented version each step
    stems_aug = {s: xa[:, 2 * i:2 * i + 2] for i, s in enumerate(("vocals", "bass", "drums", "other"))} if a.aug else None

    stager = None
    if a.ingest != "resident":   # PCIe-inclusive variant: pinned host batches -> async H2D -> kernels
        from mst_amd import ingest
        hx = x.cpu()
        hx = ingest.float_to_pcm16(hx) if a.ingest == "pcm16" else hx
        host_batches = [hx.pin_memory(), torch.roll(hx, 3, 0).pin_memory()]   # what DataLoader(pin_memory=True) yields
        stager = ingest.DeviceStager(tuple(hx.shape), hx.dtype, dev)
        state = {"fut": stager.submit(host_batches[0]), "k": 0}
        del hx

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()   # materialise the hipEvent_t so the raw handle can be passed through the C ABI
        return e

    from mst_amd import _lib as mlib

    def layout_now():   # the layout the encoder's CURRENT precision mode reads fastest, if stage A can write it
        if backend != "hip" or a.logmel_layout == "ref":
            return mlib.LOGMEL_REF
        lay = model.hip_encoder().preferred_layout()
        return lay if fe.plan().supports_layout(lay) else mlib.LOGMEL_REF

    marks, pending = [], []
    pool = [[ev() for _ in range(12)] for _ in range(a.steps)]   # events are created outside the timed region

    def step(timed):
        with torch.no_grad():
            evs = pool[len(marks)] if timed else None
            e0, e1, e2 = evs[:3] if timed else (None, None, None)
            crit.gather_events = (evs[9], evs[10]) if timed and world > 1 else None
            lay = layout_now()
            if timed:
                e0.record()
            if a.aug:   # negatives = degraded anchors (README triplet design): clips 2, 5, 8, ... of the batch
                # xa[neg] = augment(x[neg]): the reference's `.clone()` (src/mixing_utils.py:386) is the chain's own first read
                augm.augment_packed_(xa[2::3], decisions=pending.pop() if pending else None, src=x[2::3])
                feats, logmel = fe.features_and_logmel(stems_aug, lay, lay == mlib.LOGMEL_CM16)
                last["stems"] = stems_aug
            elif stager is not None:
                fut = state["fut"]
                xin = fut.get()
                state["k"] += 1
                state["fut"] = stager.submit(host_batches[state["k"] % 2])   # next batch's H2D overlaps this step
                feats, logmel = fe.features_and_logmel(ingest.stems_views(xin), lay, lay == mlib.LOGMEL_CM16)
                stager.release(fut)
                last["stems"] = None
            else:
                feats, logmel = fe.features_and_logmel(stems, lay, lay == mlib.LOGMEL_CM16)
                last["stems"] = stems
            if timed:
                e1.record()
            if backend == "hip":
                kev = evs[3:9] if timed else None
                emb = model.hip_encoder().forward(logmel, feats, events=kev)
            else:
                kev = None
                emb = model.forward_from_logmel(logmel, feats)
            if timed:
                e2.record()
            if a.aug:   # host RNG work for the NEXT step overlaps this step's kernels (a data-loader worker's job)
                pending.append(augm.draw_decisions(B // 3))
            loss = crit(emb, labels)
            if timed:
                evs[11].record()
                marks.append((e0, e1, e2, kev, evs[11], crit.gather_events))
            last["emb"] = emb
            return loss

    if a.train:   # training step: same data, same metric unit; reported with its own workload string
        model.train()
        model.train_backend = a.train_backend   # "hip" raises when the hand-written trunk cannot take the call: no library fallback is ever timed
        model.train_precision = {"fp32": "fp32", "f16": "f16", "f16x3": "f16x3", "amp": "auto"}[a.train_precision]
        model.sync_bn = bool(a.sync_bn)
        try:   # the multi-tensor ("fused") AdamW of PyTorch: the same update in a handful of launches instead of ~25
            opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
        except (TypeError, RuntimeError):
            opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
        crit_t = InfoNCELoss(0.1, gather=world > 1, check="deferred")
        amp = a.train_precision == "amp"
        scaler = torch.amp.GradScaler("cuda") if amp else None
        reducer = None
        if world > 1:
            from mst_amd.dist import GradientReducer
            reducer = GradientReducer(model)

        # the reference trainer's call (src/train.py:253): the Dataset's deferred feature rows + the stems; ONE stage-A launch
        # inside model.forward yields the features and the log-mel in the layout the training trunk of this precision reads
        from mst_amd.mixing_utils import deferred_features
        deferred = torch.stack([deferred_features(model.film_encoder.feature_dim)] * B).to(dev)

        def train_step():
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                emb = model(stems, deferred)
                loss = crit_t(emb, labels)
            opt.zero_grad(set_to_none=True)
            (scaler.scale(loss) if amp else loss).backward()
            if reducer is not None:   # data-parallel: the parameter gradients are summed over the ranks (see loss.InfoNCELoss) by
                reducer.wait()        # four bucketed all-reduces that were launched DURING the backward pass (mst_amd/dist.py)
            if amp:
                scaler.step(opt)
                scaler.update()
            else:
                opt.step()
            return loss
        for _ in range(max(a.warmup, 5)):
            train_step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = train_step()
        crit_t.finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({
                "metric": "triplets/sec (10 s @ 44.1 kHz, 4-stem, bs=24)", "value": round(world * a.triplets * a.steps / t.item(), 3),
                "unit": "triplets/s", "n_gpus": world, "world": world, "rccl_ranks_seen": ranks_seen, "steps": a.steps,
                "warmup": max(a.warmup, 5),
                "ms_per_step": round(t.item() / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": {"fp32": "f32", "f16": "f16 operands / f32 accumulate (conv trunk)",
                                               "f16x3": "f32-equivalent (conv trunk on 3-term split-precision f16 MFMA, f32 accumulate)",
                                               "amp": "autocast f16 + GradScaler"}[a.train_precision], "data": "synthetic",
                "config": {"workload": f"NOT THE CONTRACT LINE -- full TRAINING step ({a.train_backend} encoder backend, "
                                       f"precision {a.train_precision}): HIP stage A, "
                                       "encoder forward + backward (train-mode BatchNorm, Dropout 0.3), InfoNCE forward + backward, "
                                       f"AdamW; {a.triplets} triplets = {B} clips of {a.seconds:.0f} s per GPU",
                           "clips_per_gpu": B, "train_backend": a.train_backend, "train_precision": a.train_precision,
                           "sync_bn": bool(a.sync_bn),
                           "loss": float(loss.detach()),
                           "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    for _ in range(a.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step(True)
    crit.finish()   # the last step's guard (inside the timed region: a device -> host read of one word)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0

    # per-rank figures for the N > 1 line (why the scaling factor is what it is): every rank's own wall time for the K steps
    per_rank_ms = None
    if world > 1:
        mine = torch.tensor([el / a.steps * 1e3], device=dev, dtype=torch.float64)
        allr = torch.empty(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allr, mine)
        per_rank_ms = [round(v, 4) for v in allr.tolist()]
    t = torch.tensor([el], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = t.item()

    # post-timing check of the timed workload itself (outside the timed region): batch independence on every rank; the oracle
    # comparison rides on the cpu_baseline leg (rank 0, N = 1), see verify()
    main_emb = last["emb"].clone()
    oracle_clips = [0, (B // 2) // 3 * 3, B - 1] if B >= 3 else list(range(B))   # first, a middle anchor, the last (a negative under --aug)
    verified, base, ox = None, None, None
    if not a.no_verify:
        if rank == 0 and world == 1 and not a.no_cpu_baseline and last["stems"] is not None:
            ox = torch.cat([torch.cat([last["stems"][s][c:c + 1] for s in ("vocals", "bass", "drums", "other")], 1)
                            for c in oracle_clips], 0).float().cpu()
        verified = verify(model, fe, layout_now, backend, last, a.verify_b1)
        okf = torch.tensor([1.0 if verified["ok"] else 0.0], device=dev)
        if world > 1:
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        verified["ok_all_ranks"] = bool(okf.item() == 1.0)

    # opt-in precision modes, measured after (outside) the contract's timed region; the headline stays exact fp32
    alt = None
    main_marks = len(marks)
    if backend == "hip" and a.precision == "fp32" and not a.aug and a.ingest == "resident" and a.config == "default":
        notes = {"f16x3-all": "conv1+conv2 on f16 MFMA with 3-term split precision, fp32 accumulate; same 1e-4 parity tests as the exact-fp32 kernels",
                 "f16": "conv1+conv2 with plain f16 operands, fp32 accumulate (the reference's --use_amp conv arithmetic); "
                        "NOT within the 1e-4 parity bar, shown for headroom only"}
        alt = []
        n_alt = max(5, a.steps // 2)
        for mode, note in notes.items():
            model.conv1_precision = mode
            try:
                for _ in range(3):
                    step(False)
                pool.extend([[ev() for _ in range(12)] for _ in range(n_alt)])
                m0 = len(marks)
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize()
                ta = time.perf_counter()
                for _ in range(n_alt):
                    step(True)
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize()
                tb = torch.tensor([time.perf_counter() - ta], device=dev, dtype=torch.float64)
                if world > 1:
                    dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                mm = marks[m0:]
                k_ms = [sum(m[3][i].elapsed_time(m[3][i + 1]) for m in mm) / len(mm) for i in range(5)]
                a_ms = sum(m[0].elapsed_time(m[1]) for m in mm) / len(mm)
                terms = 3 if mode.startswith("f16x3") else 1
                ns_, nfr = model.audio_encoder.n_subbands, 1 + T // hop
                fl1 = B * ns_ * 2.0 * 32 * 392 * split * nfr
                fl2 = B * ns_ * 2.0 * 64 * 1568 * 8 * (nfr // 5)
                d = (last["emb"] - main_emb).abs()
                row = {"mode": f"{mode}: {note}", "value": round(world * a.triplets * n_alt / tb.item(), 3),
                       "unit": "triplets/s", "ms_per_step": round(tb.item() / n_alt * 1e3, 4), "steps": n_alt,
                       "kernels_ms": {"stage_a": round(a_ms, 4), "film_mlp": round(k_ms[0], 4), "conv1": round(k_ms[1], 4),
                                      "conv2": round(k_ms[2], 4), "attn_scores": round(k_ms[3], 4), "attn_pool_proj": round(k_ms[4], 4)},
                       # executed MFMA flops (each product as `terms` f16 MFMAs) against the dense f16 peak; the chip holds ~1.8-1.9 GHz
                       # under f16 MFMA load, i.e. ~1.9 PFLOP/s is what a perfect kernel would reach
                       "roofline": {"bound": "mfma", "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "mfma_terms": terms,
                                    "conv1_achieved_executed": round(terms * fl1 / (k_ms[1] * 1e-3) / 1e12, 1),
                                    "conv1_frac": round(terms * fl1 / (k_ms[1] * 1e-3) / 1e12 / MFMA_F16_PEAK_TF, 4),
                                    "sustained_on_random_operands": MFMA_F16_SUSTAINED_TF,
                                    "conv2_achieved_executed": round(terms * fl2 / (k_ms[2] * 1e-3) / 1e12, 1),
                                    "conv2_frac": round(terms * fl2 / (k_ms[2] * 1e-3) / 1e12 / MFMA_F16_PEAK_TF, 4)},
                       "embedding_error_vs_fp32_kernels": {
                           "normwise": float((d.amax(dim=1) / main_emb.abs().amax(dim=1)).max()),
                           "max_rel_elementwise": float((d / torch.maximum(main_emb.abs(), 1e-2 * main_emb.abs().amax(dim=1, keepdim=True))).max())}}
                if not a.no_verify:
                    v = verify(model, fe, layout_now, backend, last, a.verify_b1)
                    row["verified"] = v
                    if not v["ok"]:
                        verified["ok"] = False
                alt.append(row)
            except Exception as ex:  # a mode that refuses to run is reported, not hidden
                alt.append({"mode": mode, "error": str(ex)[:200]})
        model.conv1_precision = a.precision
        del marks[main_marks:]

    msA = sum(m[0].elapsed_time(m[1]) for m in marks) / len(marks)
    msB = sum(m[1].elapsed_time(m[2]) for m in marks) / len(marks)
    # per-step periods on the GPU's own clock (step i's first event to step i+1's; the last step: to its loss event), for the
    # median SURVEY 8(d) asks for next to the mean over the wall clock
    periods = sorted([marks[i][0].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)] + [marks[-1][0].elapsed_time(marks[-1][4])])
    ms_median = periods[len(periods) // 2] if len(periods) % 2 else 0.5 * (periods[len(periods) // 2 - 1] + periods[len(periods) // 2])
    multi = None
    if world > 1:   # what explains the N-GPU scaling factor: each rank's own kernels (stage A + B) and the exchange's share
        own = msA + msB
        gat = sum(m[5][0].elapsed_time(m[5][1]) for m in marks) / len(marks)
        mine = torch.tensor([own, gat, sum(m[2].elapsed_time(m[4]) for m in marks) / len(marks)], device=dev, dtype=torch.float64)
        allr = torch.empty(world * 3, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(world, 3).cpu()
        multi = {"per_rank_ms_per_step_wall": per_rank_ms,
                 "per_rank_own_kernels_ms": {"min": round(float(allr[:, 0].min()), 4), "max": round(float(allr[:, 0].max()), 4)},
                 "all_gather_ms": {"min": round(float(allr[:, 1].min()), 4), "max": round(float(allr[:, 1].max()), 4),
                                   "note": "events around the packed embedding + label all-gather on the compute stream: includes waiting for the slowest rank's embeddings"},
                 "exchange_plus_loss_ms": {"min": round(float(allr[:, 2].min()), 4), "max": round(float(allr[:, 2].max()), 4)},
                 "all_gather_share_of_step": round(float(allr[:, 1].max()) / (el / a.steps * 1e3), 4)}
    if rank == 0:
        n_frames = 1 + T // hop
        bytes_a = B * (8 * T * 4 + 8 * n_mels * n_frames * 4 + 64 * 4)      # 21,169,664 B/clip
        bytes_fused = B * (8 * T * 4 + 64 * 4 + embed * 4)                   # 14,115,584 B/clip (waveform in,
        #                                                                      features + embedding out)
        W1 = n_frames // 5
        ns = model.audio_encoder.n_subbands
        sub = max(1, split // 10)
        H1 = split // sub
        rows2 = (H1 // 4) * 4                                               # conv2 rows that reach MaxPool(4,4)
        Cp = 64 * ns * (H1 // 4)
        flops_c1 = B * ns * 2.0 * 32 * 392 * split * n_frames               # 9.510 GFLOP/clip at the default
        flops_c2_alg = B * ns * 2.0 * 64 * 1568 * H1 * W1                   # 7.595 GFLOP/clip: what the reference computes
        flops_c2_exec = B * ns * 2.0 * 64 * 1568 * rows2 * W1               # HIP path: rows 8, 9 of 10 never reach MaxPool(4,4)
        flops_head = B * (2.0 * Cp * 256 * (W1 // 4) + 2.0 * Cp * embed + 2.0 * (64 * 256 + 256 * 256 + 256 * ns * 192))
        flops_b_alg = flops_c1 + flops_c2_alg + flops_head                  # 17.17 GFLOP/clip at the default
        # the exact-fp32 HIP kernels leave out products with the zero padding above / below a plane (csrc/encoder.hip): conv1's
        # first and last tile row (2 rows each) skip 2 of the 7 tap rows; conv2's first row pair of a plane skips 2 of 7 tap rows
        flops_c1_exec = flops_c1
        if backend == "hip" and a.precision == "fp32" and sub <= 2 and split >= 4:
            flops_c1_exec = flops_c1 * (1.0 - (2.0 / (split // 2)) * (2.0 / 7.0))
            flops_c2_exec *= 1.0 - (1.0 / max(1, rows2 // 8)) * 0.25 * (2.0 / 7.0)
        flops_b_exec = flops_c1_exec + (flops_c2_exec if backend == "hip" else flops_c2_alg) + flops_head
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if backend == "hip":
            kms = [sum(m[3][i].elapsed_time(m[3][i + 1]) for m in marks) / len(marks) for i in range(5)]
            roof = {"kernel": "conv1_resident_kernel: Conv7x7(8->32)+BN+FiLM+ReLU+MaxPool(2,5), fp32 MFMA 16x16x4 implicit GEMM",
                    "bound": "mfma", "achieved": round(flops_c1 / (kms[1] * 1e-3) / 1e12, 3), "peak": MFMA_F32_PEAK_TF,
                    "unit": "TFLOP/s"}
            roof["kernels_ms"] = {"film_mlp": round(kms[0], 4), "conv1": round(kms[1], 4), "conv2": round(kms[2], 4),
                                  "attn_scores": round(kms[3], 4), "attn_pool_proj": round(kms[4], 4)}
            # `achieved` / `frac` are the ALGORITHMIC rate (the reference's flop count over this kernel's time); the MFMAs
            # actually issued are fewer -- products with zero padding are left out -- and THEIR rate is what the pipe's peak bounds
            roof["conv1_tflops_executed"] = round(flops_c1_exec / (kms[1] * 1e-3) / 1e12, 3)
            roof["frac_executed"] = round(flops_c1_exec / (kms[1] * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)
            roof["conv2_tflops_executed"] = round(flops_c2_exec / (kms[2] * 1e-3) / 1e12, 3)
            roof["conv2_tflops_algorithmic"] = round(flops_c2_alg / (kms[2] * 1e-3) / 1e12, 3)
            if os.path.exists(tp) and a.config == "default" and a.precision == "fp32":
                try:   # PMC counters cannot be collected inside this run: the committed per-launch figure of the same
                    tj = json.load(open(tp))   # kernel at the same shapes is quoted, with its source
                    traffic = tj.get("conv1", {}).get("hbm_bytes_per_launch")
                    traffic_src = f"profiles/traffic.json ({tj.get('passes', 'rocprofv3 --pmc')}); not measured by this run"
                    sa = tj.get("stage_a", {})
                    if sa:   # what limits stage A (it is issue / latency-bound, not HBM-bound): counters of the same kernel at the same shapes
                        roof["stage_a_traffic"] = sa.get("hbm_bytes_per_launch")
                        iss = sa.get("issue", {})
                        roof["stage_a_valu_frac"] = iss.get("valu_busy_frac")   # share of SIMD time the vector ALUs execute
                        roof["stage_a_lds_frac"] = iss.get("lds_busy_frac")     # share of time the CUs' LDS pipes are busy
                        roof["stage_a_wave_wait_frac"] = iss.get("wave_wait_frac")
                        roof["stage_a_counters_source"] = iss.get("source")
                except Exception:
                    traffic = None
        else:   # BASELINE configs[1]: stage B on PyTorch-ROCm library kernels -- no per-kernel events, whole-stage rate
            roof = {"kernel": "stage B on PyTorch-ROCm/MIOpen library kernels (whole stage; no hand-written kernel to time)",
                    "bound": "mfma", "achieved": round(flops_b_exec / (msB * 1e-3) / 1e12, 3), "peak": MFMA_F32_PEAK_TF,
                    "unit": "TFLOP/s"}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        roof["traffic"] = traffic
        roof["traffic_source"] = traffic_src
        roof["stage_a_ms"] = round(msA, 4)
        roof["stage_a_gbs"] = round(bytes_a / (msA * 1e-3) / 1e9, 1)
        roof["stage_a_hbm_frac"] = round(bytes_a / (msA * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        roof["stage_b_ms"] = round(msB, 4)
        roof["stage_b_tflops"] = round(flops_b_exec / (msB * 1e-3) / 1e12, 3)            # EXECUTED flops
        roof["stage_b_tflops_algorithmic"] = round(flops_b_alg / (msB * 1e-3) / 1e12, 3)  # reference's 17.17 GFLOP/clip
        roof["fused_hbm_frac"] = round(bytes_fused / (el / a.steps) / 1e9 / HBM_PEAK_GBS, 5)  # SURVEY 8(d) fused figure
        out = {
            "metric": "triplets/sec (10 s @ 44.1 kHz, 4-stem, bs=24)",
            "value": round(world * a.triplets * a.steps / el, 3),
            "unit": "triplets/s",
            "n_gpus": world,
            "world": dist.get_world_size() if world > 1 else 1,
            "rccl_ranks_seen": ranks_seen,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 4),
            "ms_per_step_median": round(ms_median, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if a.precision == "fp32" else f"f32 ({a.precision}: convs on split-precision f16 MFMA, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": ("" if a.config == "default" else f"NOT THE CONTRACT SHAPE ({a.config}: n_fft {n_fft}, hop {hop}, {n_mels} mels, sub-bands {split}/{overlap}, {embed}-d) -- ") +
                       ("configs[2]" if backend == "hip" else "configs[1]") +
                       f": synthetic {a.seconds:.0f} s stereo 4-stem clips, {a.triplets} triplets = {B} clips per GPU, "
                       f"HIP STFT+{n_mels}-mel+64-d features, encoder fwd in " +
                       ("HIP (fp32 MFMA)" if backend == "hip" else "PyTorch-ROCm") +
                       ", InfoNCE on all-gathered embeddings" + (", HIP augmentation chain on the negatives" if a.aug else "") +
                       ("" if a.ingest == "resident" else f"; PCIe-INCLUSIVE: every batch staged from pinned host memory as {a.ingest}"),
                       "clips_per_gpu": B, "clip_samples": T, "n_fft": n_fft, "hop": hop, "n_mels": n_mels,
                       "encoder_backend": backend, "conv1_precision": a.precision,
                       "logmel_layout": {0: "reference (B,8,M,F)", 1: "channel-minor [B][F][M][8] fp32 (encoder-internal)",
                                         2: "channel-minor float16 hi/lo planes (encoder-internal)"}[layout_now()], "parallelism": f"clip-sharded x{world}", "loss": float(loss),
                       "loss_guard": "the reference's no-positive-pairs RuntimeError is evaluated one call late from a pinned word (InfoNCELoss check='deferred'): no per-step device->host read"},
            "roofline": roof,
        }
        out["alt"] = alt
        if multi is not None:
            out["multi_gpu"] = multi
        if world == 1 and not a.no_cpu_baseline:
            cfg = dict(sr=sr, n_fft=n_fft, hop=hop, n_mels=n_mels, split=split, overlap=overlap)
            if ox is None:   # no stems to check against (--ingest / --no-verify): the baseline runs on the synthetic clips themselves
                ox = x[oracle_clips].float().cpu()
            out["cpu_baseline"], oemb = cpu_baseline(model.state_dict(), ox, cfg)
            if verified is not None and last["stems"] is not None:
                verify_oracle(verified, main_emb, oracle_clips, oemb, a.precision)
        out["config"]["verified"] = verified if verified is not None else "skipped (--no-verify)"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if verified is not None and not (verified["ok"] and verified.get("ok_all_ranks", True)):
        print(f"bench.py: the timed workload FAILED its correctness check: {json.dumps(verified)}", file=sys.stderr)
        sys.exit(4)


if __name__ == "__main__":
    main()
