"""GPU: cross-rank BatchNorm statistics (`MixingStyleEncoder.sync_bn`) over a real torch.distributed group.
Two ranks on the box's single GPU (gloo carries the CUDA tensors here -- RCCL refuses two ranks on one device; the calls are
the ones RCCL serves on a node: all_reduce SUM on int64, MAX on int32): the summed parameter gradients of the two-rank step
equal the single-process step on the whole batch, which is what the reference trains (src/train.py:211 on one GPU).  The ranks
hold DIFFERENT numbers of clips (2 and 4).  An RCCL-backed variant runs when at least two GPUs are visible."""
import copy
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, precision, ret, backend="gloo"):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    import cases
    from oracle import mel as omel
    from test_encoder_gpu import build_model
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        cfg = cases.CFG_DEFAULT
        model, _ = build_model(cfg)
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        whole = copy.deepcopy(model)
        model.train(), whole.train()
        model.train_backend = whole.train_backend = "hip-strict"
        model.train_precision = whole.train_precision = precision
        model.sync_bn = True
        B, T = 6, 44100
        cut = [0, 2, B]   # UNEQUAL shards (2 + 4 clips, a ragged last batch): the global clip count rides with the statistics
        x = torch.stack([cases.synth_clip(c % 4, T) * (1.0 + 0.1 * c) for c in range(B)], 0).cuda()
        g = torch.Generator().manual_seed(43)
        feats = (torch.randn(B, 64, generator=g) * 2.0).cuda()
        R = torch.randn(B, cfg["embed_dim"], generator=g).cuda()
        with torch.no_grad():
            lm = model.audio_encoder.mel_preprocessor(omel.tensor_to_stems_dict(x))
        sl = slice(cut[rank], cut[rank + 1])
        from mst_amd.dist import GradientReducer
        red = GradientReducer(model)     # C2: bucketed all-reduces launched DURING the backward (the trunk launches its own two)
        loss = (model.forward_from_logmel(lm[sl].contiguous(), feats[sl]) * R[sl]).sum()
        loss.backward()
        order = red.wait()
        assert order[0] == "head" and order[-1] == "film" and order.index("conv2") < order.index("conv1"), order
        tot = loss.detach().clone()
        dist.all_reduce(tot)
        grads = {n: p.grad.clone() for n, p in model.named_parameters()}
        if rank == 0:
            lw = (whole.forward_from_logmel(lm, feats) * R).sum()
            lw.backward()
            worst = (0.0, "")
            for n, p in whole.named_parameters():
                den = p.grad.abs().max().item()
                if den > 1e-9 and not n.endswith(("conv1.bias", "conv2.bias", "attention.2.bias")):
                    worst = max(worst, ((grads[n] - p.grad).abs().max().item() / den, n))
            stat = max((a - b).abs().max().item() for (_, a), (_, b) in zip(model.named_buffers(), whole.named_buffers())
                       if a.dtype.is_floating_point)
            ret["loss"] = abs(tot.item() - lw.item()) / abs(lw.item())
            ret["worst"], ret["stat"] = worst, stat
        # the same step with .grad tensors that already exist (zeroed in place): autograd accumulates into them instead of adopting
        # the trunk's stacked views, and the reducer must still leave the reduced gradients there (it used to leave the local ones)
        model.zero_grad(set_to_none=False)
        (model.forward_from_logmel(lm[sl].contiguous(), feats[sl]) * R[sl]).sum().backward()
        red.wait()
        ret[f"inplace_equal_{rank}"] = all(torch.equal(p.grad, grads[n]) for n, p in model.named_parameters())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["fp32", "f16"])
def test_two_rank_step_with_sync_bn_equals_the_single_process_step(precision):
    import socket
    with socket.socket() as so:   # a free port, as bench.launch_ranks does
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, precision, ret), nprocs=2, join=True)
    assert {"loss", "worst", "stat"} <= set(ret.keys()), f"rank 0 did not report (a worker failed before the comparison): {dict(ret)}"
    tol = 1e-4 if precision == "fp32" else 3e-2   # f16: per-rank range scales move single float16 roundings
    print(f"sync_bn, 2 ranks, {precision}: loss {ret['loss']:.2e}, worst gradient {ret['worst']}, running statistics {ret['stat']:.2e}")
    assert ret["loss"] < (1e-5 if precision == "fp32" else 1e-3) and ret["worst"][0] < tol and ret["stat"] < 1e-4, dict(ret)
    assert ret.get("inplace_equal_0") and ret.get("inplace_equal_1"), "zero_grad(set_to_none=False): reduced gradients must be bit-equal to the first step's"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank; this box has one")
def test_two_rank_step_with_sync_bn_over_rccl():
    """The same step with the statistics exchanged by RCCL (backend "nccl"), one GPU per rank."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, "fp32", ret, "nccl"), nprocs=2, join=True)
    assert {"loss", "worst", "stat"} <= set(ret.keys()), dict(ret)
    assert ret["loss"] < 1e-5 and ret["worst"][0] < 1e-4 and ret["stat"] < 1e-4, dict(ret)
