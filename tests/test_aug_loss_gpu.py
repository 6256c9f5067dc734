"""GPU parity: augmentation chain (decisions bit-exact, audio within tolerance) and InfoNCE kernel vs oracle/goldens."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import augment as oaug
from oracle import loss as oloss
from oracle import mel as omel

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def close(a, ref, rtol=1e-4, atol=2e-5):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(ref, np.float64), rtol=rtol, atol=atol)


def log_rand():
    real, draws = torch.rand, []

    def rand(*a, **k):
        v = real(*a, **k)
        draws.append(float(v.flatten()[0]))
        return v
    return real, rand, draws


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 11])
def test_augment_stems_vs_golden_and_oracle(seed):
    from mst_amd.mixing_utils import AudioAugmenter
    g = np.load(os.path.join(G, "augment.npz"))
    x = cases.feature_case("synth1", 33075)
    aug = AudioAugmenter(44100, 9.0, 0.5)
    real, rand, draws = log_rand()
    torch.manual_seed(seed)
    torch.rand = rand
    try:
        y = aug.augment_stems({k: v.cuda() for k, v in omel.tensor_to_stems_dict(x).items()})
    finally:
        torch.rand = real
    torch.cuda.synchronize()
    # decisions: the RNG consumption is bit-identical to the reference's
    assert np.array_equal(np.array(draws, dtype=np.float64), g[f"seed{seed}.rand_draws"])
    assert int(g[f"seed{seed}.reverb"]) == int("reverb_ir" in aug.last_trace)
    if "reverb_ir" in aug.last_trace:   # the impulse response is the reference's draw, bit for bit (src/mixing_utils.py:460-464)
        r = aug.last_trace["reverb_randn"]
        assert np.array_equal(r[:8].numpy(), g[f"seed{seed}.ir_randn_head"])
        assert float(r.double().sum()) == float(g[f"seed{seed}.ir_randn_sum"])
        t = torch.linspace(0, 0.5, 22050)
        assert torch.equal(aug.last_trace["reverb_ir"], torch.exp(-t / 0.125) * r * 0.1)
    y8 = omel.stems_dict_to_tensor({k: v.cpu() for k, v in y.items()})
    idx = torch.from_numpy(g[f"seed{seed}.idx"])
    close(y8.flatten()[idx], g[f"seed{seed}.samples"])
    close(y8.double().sum(-1), g[f"seed{seed}.chan_sum"], rtol=1e-4, atol=2e-2)
    close((y8.double() ** 2).sum(-1), g[f"seed{seed}.chan_sqsum"], rtol=2e-4, atol=1e-5)
    # full tensor against the oracle run with the same seed
    torch.manual_seed(seed)
    ref, _ = oaug.augment_stems(omel.tensor_to_stems_dict(x))
    close(y8, omel.stems_dict_to_tensor(ref))
    # inputs are not mutated (the reference clones)
    assert torch.equal(x, cases.feature_case("synth1", 33075))


def test_batched_full_size_matches_sequential_oracle():
    """(B,2,T) dict at the BASELINE clip length: clip b gets the decisions the b-th sequential reference call would."""
    from mst_amd.mixing_utils import AudioAugmenter
    T = 441000
    x = torch.stack([cases.synth_clip(c, T) for c in (0, 1, 2)], 0)
    torch.manual_seed(7)
    aug = AudioAugmenter()
    y = aug.augment_stems({k: v.cuda() for k, v in omel.tensor_to_stems_dict(x).items()})
    y8 = omel.stems_dict_to_tensor({k: v.cpu() for k, v in y.items()})
    torch.manual_seed(7)
    for b in range(3):
        ref, tr = oaug.augment_stems(omel.tensor_to_stems_dict(x[b]))
        assert ("reverb_ir" in tr) == ("reverb_ir" in aug.last_trace[b])
        close(y8[b], omel.stems_dict_to_tensor(ref))


def test_single_effects():
    from mst_amd.mixing_utils import AudioAugmenter
    g = np.load(os.path.join(G, "augment.npz"))
    aug = AudioAugmenter()
    x = cases.feature_case("synth1", 33075)[4:6]
    close(aug.apply_compression(x.cuda()).cpu()[:, :4096], g["compress.samples"], rtol=1e-5, atol=1e-7)
    torch.manual_seed(123)
    close(aug.apply_reverb(x.cuda()).cpu()[:, :4096], g["reverb.out_head"], rtol=1e-4, atol=1e-5)
    torch.manual_seed(9)
    close(aug.apply_spectral_tilt(x.cuda()).cpu()[:, :4096], g["tilt.out_head"], rtol=1e-5, atol=1e-6)
    torch.manual_seed(10)
    close(aug.apply_bandwidth_limit(x.cuda()).cpu()[:, :4096], g["bw.out_head"], rtol=1e-5, atol=1e-6)


def test_parametric_compressor_matches_the_reference_at_non_default_settings():
    """`apply_compression(audio, threshold, ratio)` (src/mixing_utils.py:435-447) is parametric in the reference; the fixture
    holds its output at three non-default settings (tests/golden/make_golden.py::gen_compress_param), on a signal with exact zeros,
    both signs and samples on both sides of every threshold.  Also against the oracle, and the default setting through the general
    path equals the closed-form path to 1e-6."""
    from mst_amd.mixing_utils import AudioAugmenter
    from oracle import augment as oaug
    g = np.load(os.path.join(G, "augment_compress_param.npz"))
    aug = AudioAugmenter()
    x = cases.feature_case("synth1", 33075)[4:6].clone()
    x[:, :64] = 0.0
    x[:, 64:128] *= 8.0
    assert abs(float(x.double().abs().sum()) - float(g["input_checksum"])) < 1e-9 * float(g["input_checksum"]), "fixture input differs"
    for tag in ("m12_2", "m30_8", "m6_1p5"):
        thr, ratio = (float(v) for v in g[f"{tag}.params"])
        y = aug.apply_compression(x.cuda(), threshold=thr, ratio=ratio).cpu()
        close(y[:, :8192], g[f"{tag}.samples"], rtol=1e-5, atol=1e-7)
        close(y, oaug.compress(x, thr, ratio), rtol=1e-5, atol=1e-7)
        assert torch.equal(y[:, :64], torch.zeros(2, 64)), "sign(0) * anything = 0"
    a = aug.apply_compression(x.cuda()).cpu()
    b = aug.apply_compression(x.cuda(), threshold=-20.0, ratio=4.000001).cpu()   # forces the general path at (almost) the default
    close(b, a, rtol=2e-6, atol=1e-7)
    with pytest.raises(ValueError):
        aug.apply_compression(x.cuda(), ratio=0)


def test_no_decision_is_identity():
    from mst_amd.mixing_utils import AudioAugmenter
    x = cases.feature_case("white", 20000)
    y = AudioAugmenter(prob=0.0).augment_stems({k: v.cuda() for k, v in omel.tensor_to_stems_dict(x).items()})
    assert torch.equal(omel.stems_dict_to_tensor({k: v.cpu() for k, v in y.items()}), x)


def test_infonce_kernel_vs_golden_and_oracle():
    from mst_amd.loss import InfoNCELoss, info_nce_rows, info_nce_rows_hip
    g = np.load(os.path.join(G, "infonce.npz"))
    gen = torch.Generator().manual_seed(5)
    for tag, n, d, nsong in (("pairs48", 48, 768, 24), ("gathered384", 384, 768, 192), ("triples", 12, 16, 4)):
        e = torch.randn(n, d, generator=gen)
        lab = torch.arange(n) % nsong
        loss = InfoNCELoss(0.1)(e.cuda(), lab.cuda())
        close(loss.item(), g[f"{tag}.loss"], rtol=1e-5, atol=1e-6)
        close(loss.item(), oloss.info_nce(e, lab, 0.1).item(), rtol=1e-5, atol=1e-6)
        # sharded rows: sum over shards == whole
        s_all, c_all = info_nce_rows_hip(e.cuda(), lab.cuda(), 0, n, 0.1)
        s1, c1 = info_nce_rows_hip(e.cuda(), lab.cuda(), 0, n // 2, 0.1)
        s2, c2 = info_nce_rows_hip(e.cuda(), lab.cuda(), n // 2, n - n // 2, 0.1)
        close((s1 + s2).item(), s_all.item(), rtol=1e-5)
        assert (c1 + c2).item() == c_all.item() == n
        st, ct = info_nce_rows(e.cuda(), lab.cuda(), 0, n, 0.1)
        close(s_all.item(), st.item(), rtol=1e-5)
    with pytest.raises(RuntimeError):
        InfoNCELoss(0.1)(torch.randn(4, 8).cuda(), torch.arange(4).cuda())


def test_infonce_backward_vs_autograd_of_the_oracle():
    """`mst_infonce_backward` == torch autograd through the oracle's loss (float64), whole batch and
    row-sharded (each shard yields gradients for ALL rows; their sum is the whole-batch gradient)."""
    from mst_amd.loss import InfoNCELoss, info_nce_rows_hip
    gen = torch.Generator().manual_seed(11)
    for n, d, nsong in ((48, 768, 24), (12, 16, 4), (9, 32, 5)):       # last: songs 4 has a single clip -> no positive
        e = torch.randn(n, d, generator=gen) * (1.0 + torch.rand(n, 1, generator=gen))
        lab = torch.arange(n) % nsong
        ed = e.double().requires_grad_(True)
        lo = oloss.info_nce(ed, lab, 0.1)
        lo.backward()
        ec = e.cuda().requires_grad_(True)
        lg = InfoNCELoss(0.1)(ec, lab.cuda())
        lg.backward()
        close(lg.item(), lo.item(), rtol=1e-5, atol=1e-6)
        ref = ed.grad.float()
        assert (ec.grad.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-7
        # sharded rows
        nvalid = int(sum((lab == lab[i]).sum() > 1 for i in range(n)))
        tot = torch.zeros_like(ref)
        for r0, rows in ((0, n // 3), (n // 3, n - n // 3)):
            es = e.cuda().requires_grad_(True)
            s, c = info_nce_rows_hip(es, lab.cuda(), r0, rows, 0.1)
            (s / nvalid).backward()
            tot += es.grad.cpu()
        assert (tot - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-7


def test_packed_strided_in_place_variant_equals_augment_stems():
    """`augment_packed_` on every third clip of a batch tensor (clip stride 3 * 8 * T; `mst_aug_apply_strided`) == `augment_stems`
    on those clips with the same decisions, bit for bit; the clips in between are untouched.  T % 4 == 0: the LDS-slab
    passes and the 16-byte redistribution; T % 4 != 0: the chunk-walk kernels."""
    from mst_amd.mixing_utils import AudioAugmenter
    for T in (44100, 33075):
        x = torch.stack([cases.synth_clip(c, T) for c in range(7)], 0).cuda()
        aug = AudioAugmenter()
        torch.manual_seed(21)
        dec = aug.draw_decisions(3)
        ref = aug.augment_stems(omel.tensor_to_stems_dict(x[0::3]), decisions=dec)
        ref8 = omel.stems_dict_to_tensor(ref)
        y = x.clone()
        aug.augment_packed_(y[0::3], decisions=dec)
        torch.cuda.synchronize()
        assert torch.equal(y[0::3], ref8)
        for b in (1, 2, 4, 5):
            assert torch.equal(y[b], x[b])
        assert not torch.equal(y[0], x[0])


def test_out_of_place_variant_equals_clone_then_in_place():
    """`augment_packed_(dst, src=...)` (`mst_aug_apply_from`): dst = augment(src) bit-equal to the reference's order -- clone, then
    augment the clone (src/mixing_utils.py:386) -- with src untouched; a clip WITHOUT any decision is copied; overlapping tensors
    are refused."""
    from mst_amd import _lib
    from mst_amd.mixing_utils import AudioAugmenter
    for T in (44100, 33075):
        x = torch.stack([cases.synth_clip(c, T) for c in range(7)], 0).cuda()
        keep = x.clone()
        aug = AudioAugmenter()
        torch.manual_seed(21)
        dec = aug.draw_decisions(3)
        ref = x.clone()
        aug.augment_packed_(ref[0::3], decisions=dec)
        y = torch.full_like(x, float("nan"))
        aug.augment_packed_(y[0::3], decisions=dec, src=x[0::3])
        torch.cuda.synchronize()
        assert torch.equal(x, keep), "src was written"
        assert torch.equal(y[0::3], ref[0::3])
        assert bool(torch.isnan(y[1]).all()), "clips between the strided ones were touched"
        # no decision at all: a plain copy
        clips = (_lib.AugClip * 2)()
        for b in range(2):
            for i in range(4):
                clips[b].stem[i].gain = 1.0
        z = torch.full((2, 8, T), float("nan"), device="cuda")
        aug.augment_packed_(z, decisions=(clips, [None, None], [{}, {}]), src=x[1:3])
        torch.cuda.synchronize()
        assert torch.equal(z, x[1:3])
    with pytest.raises(_lib.MstError, match="overlap"):
        aug.augment_packed_(x[1:3], decisions=(clips, [None, None], [{}, {}]), src=x[0:2])


def test_infonce_deferred_guard_raises_one_call_later_and_corrupts_nothing():
    """`InfoNCELoss(check="deferred")`: the reference's "No positive pairs found in batch!" RuntimeError (src/loss.py) without a
    device -> host read per call -- a batch without positives gives a ZERO loss with zero gradients, and the error is raised by the
    next call (or `finish()`); batches with positives give exactly the "sync" value."""
    from mst_amd.loss import InfoNCELoss
    g = torch.Generator().manual_seed(2)
    e = torch.randn(8, 64, generator=g).cuda().requires_grad_(True)
    good, bad = (torch.arange(8) // 2).cuda(), torch.arange(8).cuda()
    sync, lazy = InfoNCELoss(0.1), InfoNCELoss(0.1, check="deferred")
    assert torch.equal(sync(e, good), lazy(e, good))
    with pytest.raises(RuntimeError, match="No positive pairs"):
        sync(e, bad)
    loss = lazy(e, bad)                        # no error yet
    loss.backward()
    assert loss.item() == 0.0 and float(e.grad.abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="No positive pairs"):
        lazy(e, good)                          # the previous call's guard
    assert torch.equal(lazy(e, good), sync(e, good))   # the error was consumed; the criterion works on
    lazy(e, bad)
    with pytest.raises(RuntimeError, match="No positive pairs"):
        lazy.finish()
    lazy.finish()                              # nothing pending
