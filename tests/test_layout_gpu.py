"""GPU parity of the encoder-internal CHANNEL-MINOR log-mel layouts (include/mst.h MST_LOGMEL_CM32 / CM16).

Stage A writes the same log-mel VALUES whatever the layout (the arithmetic up to the store is shared), and conv1 stages the
same LDS patch from either, so every comparison here is BIT-EXACT:
  * CM32 == the reference-layout tensor, permuted; CM16 hi / lo == the float16 split the f16 conv1 kernel used to derive
    itself (hi = f16(x), lo = f16(x - hi)); absmax == max |log-mel| per clip;
  * the eval encoder gives identical embeddings / taps from a channel-minor log-mel and from the reference layout, in
    every precision mode -- so all the golden / oracle parity tests of test_encoder_gpu.py carry over unchanged;
  * the module's forward (which negotiates the layout) equals the explicit reference-layout pipeline.
The reference computes this tensor at src/model.py:41-67 (layout (B, 8, n_mels, frames)).
"""
import numpy as np
import pytest
import torch

import cases
from oracle import mel as omel

pytestmark = pytest.mark.gpu


def _plan(n_fft=1024, hop=256, n_mels=128, bins=0):
    from mst_amd.mixing_utils import MelFeatPlan
    return MelFeatPlan(44100, n_fft, hop, n_mels, bins)


def _stems(B, T, seed=0):
    x = torch.stack([cases.synth_clip(seed + c, T) for c in range(B)], 0)
    return omel.tensor_to_stems_dict(x.cuda())


CASES = [
    ("default_10s", dict(n_fft=1024, hop=256, n_mels=128), 3, 441000),
    ("default_ragged", dict(n_fft=1024, hop=256, n_mels=128), 2, 44100 + 77),     # odd length, ragged last block
    ("default_tiny", dict(n_fft=1024, hop=256, n_mels=128), 1, 1300),              # fewer frames than one block per wave
    ("mels80", dict(n_fft=1024, hop=256, n_mels=80), 2, 66150),                    # bands past n_mels masked in the stores
    ("mels256", dict(n_fft=1024, hop=256, n_mels=256), 2, 66150),                  # 4 bands per lane (BASELINE configs[4])
    ("launcher_2048", dict(n_fft=2048, hop=512, n_mels=80), 2, 441000),            # scripts/train_baseline.sh shapes
]


@pytest.mark.parametrize("tag,cfg,B,T", CASES, ids=[c[0] for c in CASES])
def test_stage_a_channel_minor_layouts_hold_the_reference_values_bit_for_bit(tag, cfg, B, T):
    from mst_amd import _lib
    plan = _plan(**cfg)
    assert plan.supports_layout(_lib.LOGMEL_CM32) and plan.supports_layout(_lib.LOGMEL_CM16)
    d = _stems(B, T)
    ref, f_ref = plan.forward_stems(d, True, True)
    cm32, f32 = plan.forward_stems(d, True, True, _lib.LOGMEL_CM32, want_absmax=True)
    cm16, f16 = plan.forward_stems(d, True, True, _lib.LOGMEL_CM16, want_absmax=True)
    torch.cuda.synchronize()
    F = plan.frames(T)
    assert tuple(cm32.data.shape) == (B, F, cfg["n_mels"], 8) and cm32.data.dtype == torch.float32
    assert tuple(cm16.data.shape) == (B, F, cfg["n_mels"], 8) and cm16.data.dtype == torch.float16 and cm16.lo.dtype == torch.float16
    want = ref.permute(0, 3, 2, 1).contiguous()
    assert torch.equal(cm32.data, want), "CM32 must be a pure re-layout"
    hi = want.half()
    lo = (want - hi.float()).half()
    assert torch.equal(cm16.data, hi) and torch.equal(cm16.lo, lo), "CM16 = (f16(x), f16(x - f16(x)))"
    amax = ref.abs().amax(dim=(1, 2, 3))
    for lm in (cm32, cm16):
        assert torch.equal(lm.absmax.view(torch.float32), amax), "absmax = max |log-mel| per clip (float bits)"
    assert torch.equal(f32, f_ref) and torch.equal(f16, f_ref), "the features do not depend on the log-mel layout"
    assert torch.equal(cm32.to_reference(), ref)


def test_channel_minor_needs_the_sliding_window_kernels_and_says_so():
    from mst_amd import _lib
    plan = _plan(n_fft=512, hop=128, n_mels=64)   # served by the generic stage-A kernel
    assert not plan.supports_layout(_lib.LOGMEL_CM32)
    with pytest.raises(_lib.MstError, match="layout"):
        plan.forward_stems(_stems(1, 22050), True, False, _lib.LOGMEL_CM32)


def _build(cfg, precision="fp32"):
    from mst_amd.model import MixingStyleEncoder
    m = MixingStyleEncoder(channels=8, feature_dim=64, **cfg)
    sd = cases.make_state_dict(cfg, seed=42)
    full = dict(m.state_dict())
    full.update(sd)
    m.load_state_dict(full, strict=True)
    m.conv1_precision = precision
    return m.cuda().eval()


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16x3-all", "f16"])
@pytest.mark.parametrize("cfgname,T,B", [("default", 66150 + 13, 3), ("baseline_sh", 66150, 2)])
def test_encoder_is_bit_identical_from_either_layout(cfgname, T, B, precision):
    from mst_amd import _lib
    cfg = cases.CFG_DEFAULT if cfgname == "default" else cases.CFG_BASELINE_SH
    m = _build(cfg, precision)
    enc = m.hip_encoder()
    lay = enc.preferred_layout()
    assert lay == (_lib.LOGMEL_CM32 if precision == "fp32" else _lib.LOGMEL_CM16)
    plan = m.audio_encoder.mel_preprocessor.plan(0)
    d = _stems(B, T, seed=3)
    ref, feats = plan.forward_stems(d, True, True)
    cm, _ = plan.forward_stems(d, True, True, lay, want_absmax=True)
    with torch.no_grad():
        e0, t0 = enc.forward(ref, feats, taps=True)
        e1, t1 = enc.forward(cm, feats, taps=True)
    torch.cuda.synchronize()
    if precision == "fp32":   # same kernel arithmetic, same k order: bit for bit
        for k in ("film", "pool1", "pool_in"):
            assert torch.equal(t0[k], t1[k]), k
        assert torch.equal(e0, e1)
    else:   # the channel-minor f16 conv1 (conv1_f16e_kernel, bank-conflict-free) sums the SAME products -- the operands are the
        #     same bits -- over the taps in another order: fp32 accumulation-order differences only
        assert torch.equal(t0["film"], t1["film"])
        for k, a, b in (("pool1", t0["pool1"], t1["pool1"]), ("pool_in", t0["pool_in"], t1["pool_in"]), ("emb", e0, e1)):
            err = (a - b).abs().max().item() / b.abs().max().item()
            # plain f16: conv2 reads pool1 ROUNDED to float16 -- a last-bit difference of conv1's sum can land on the neighbouring
            # float16 (1e-3 of that element), which the later tensors inherit
            assert err <= (2e-6 if precision != "f16" or k == "pool1" else 1e-3), (k, err)
    # the module's own forward negotiates the layout and fills the deferred features from the same launch
    from mst_amd.mixing_utils import deferred_features
    with torch.no_grad():
        e2 = m(d, torch.stack([deferred_features(64)] * B).cuda())
    assert torch.equal(e2, e1)


def test_config5_geometry_from_channel_minor():
    """256 mels / 24 sub-bands (BASELINE configs[4] shapes), 4 bands per lane in stage A."""
    from mst_amd import _lib
    cfg = dict(sample_rate=44100, n_fft=1024, hop_length=256, n_mels=256, split_size=20, overlap=10, embed_dim=768)
    m = _build(cfg)
    plan = m.audio_encoder.mel_preprocessor.plan(0)
    d = _stems(1, 88200, seed=5)
    ref, feats = plan.forward_stems(d, True, True)
    cm, _ = plan.forward_stems(d, True, True, _lib.LOGMEL_CM32)
    with torch.no_grad():
        e0 = m.hip_encoder().forward(ref, feats)
        e1 = m.hip_encoder().forward(cm, feats)
    assert torch.equal(e0, e1)


def test_layout_that_does_not_fit_the_precision_mode_is_refused():
    from mst_amd import _lib
    m = _build(cases.CFG_DEFAULT, "fp32")
    plan = m.audio_encoder.mel_preprocessor.plan(0)
    d = _stems(1, 44100)
    cm16, feats = plan.forward_stems(d, True, True, _lib.LOGMEL_CM16, want_absmax=True)
    with pytest.raises(_lib.MstError, match="layout"):
        m.hip_encoder().forward(cm16, feats)


def test_full_size_batch_properties_channel_minor():
    """BASELINE size (72 clips of 10 s would be 0.5 GB of log-mel twice; 24 clips here): size-independent properties of the
    channel-minor path -- batch independence (a clip's values do not depend on its neighbours) and equality with the
    reference layout on a strided sample of frames."""
    from mst_amd import _lib
    plan = _plan()
    B, T = 24, 441000
    x = torch.stack([cases.synth_clip(c % 5, T) for c in range(B)], 0).cuda()
    d = omel.tensor_to_stems_dict(x)
    cm, _ = plan.forward_stems(d, True, False, _lib.LOGMEL_CM32)
    one, _ = plan.forward_stems(omel.tensor_to_stems_dict(x[7:8]), True, False, _lib.LOGMEL_CM32)
    assert torch.equal(cm.data[7], one.data[0])
    assert torch.equal(cm.data[2], cm.data[7])   # clips 2 and 7 are the same synthetic clip
    ref, _ = plan.forward_stems(omel.tensor_to_stems_dict(x[:2]), True, False)
    fr = torch.arange(0, plan.frames(T), 97, device="cuda")
    assert torch.equal(cm.data[:2].index_select(1, fr), ref.permute(0, 3, 2, 1).index_select(1, fr))
    assert np.isfinite(cm.data.float().sum().item())


@pytest.mark.parametrize("precision", ["f16", "f16x3"])
@pytest.mark.parametrize("cfgname,T,B", [("default", 44100 + 13, 5), ("baseline_sh", 66150, 3)])
def test_training_step_is_bit_identical_from_stage_a_float16_planes(cfgname, T, B, precision):
    """The float16 training trunk reads stage A's MST_LOGMEL_CM16 planes directly (conv1 forward and conv1's weight gradient:
    `mst_encoder_forward_train_in`, `mst_encoder_train_conv1_wgrad_in`).  The planes hold exactly the float16 roundings the
    reference-layout kernels derive while staging, and the arithmetic order is the same, so the whole step -- loss, every
    parameter gradient, the running statistics -- is BIT-IDENTICAL to the step from the reference-layout tensor: the training
    parity tests of test_encoder_gpu.py (which feed the reference layout) carry over.  B = 5 / 3: a ragged last clip group of
    the weight gradient's 8-clip K blocks.  The module's own forward must pick the float16 planes by itself."""
    import copy
    from mst_amd import _lib
    from test_encoder_gpu import build_model
    cfg = cases.CFG_DEFAULT if cfgname == "default" else cases.CFG_BASELINE_SH
    base, _ = build_model(cfg)
    d = _stems(B, T, seed=2)
    g = torch.Generator().manual_seed(11)
    feats = torch.randn(B, 64, generator=g).cuda()
    R = torch.randn(B, cfg["embed_dim"], generator=g).cuda()
    runs, layouts = [], []
    for how in ("reference", "planes", "module"):
        m = copy.deepcopy(base).train()
        m.train_backend, m.train_precision = "hip-strict", precision
        plan = m.audio_encoder.mel_preprocessor.plan(0)
        torch.manual_seed(77)   # the Dropout seed
        if how == "reference":
            lm, _ = plan.forward_stems(d, True, False)
            emb = m.forward_from_logmel(lm, feats)
        elif how == "planes":
            assert m._train_encoder().train_layout() == _lib.LOGMEL_CM16 and plan.supports_layout(_lib.LOGMEL_CM16)
            lm, _ = plan.forward_stems(d, True, False, _lib.LOGMEL_CM16, want_absmax=True)
            emb = m.forward_from_logmel(lm, feats)
        else:
            orig = plan.forward_stems

            def spy(*a, **k):
                layouts.append(a[3] if len(a) > 3 else k.get("layout", _lib.LOGMEL_REF))
                return orig(*a, **k)
            plan.forward_stems = spy
            emb = m(d, feats)
        loss = (emb * R).sum()
        loss.backward()
        runs.append((loss.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()},
                     {n: b.detach().clone() for n, b in m.named_buffers() if "running" in n}))
    assert layouts == [_lib.LOGMEL_CM16], layouts
    l0, g0, s0 = runs[0]
    assert np.isfinite(l0.item()) and all(torch.isfinite(v).all() for v in g0.values())
    for l1, g1, s1 in runs[1:]:
        assert torch.equal(l0, l1), (l0.item(), l1.item())
        diff = [n for n in g0 if not torch.equal(g0[n], g1[n])]
        assert not diff, f"{len(diff)} gradient tensors differ, e.g. {diff[:3]}"
        assert all(torch.equal(s0[n], s1[n]) for n in s0)


def test_fp32_training_keeps_the_reference_layout_and_converts_a_given_logmel():
    """The fp32 training kernels read the reference layout: the module does not ask stage A for planes, and a LogMel a caller
    passes anyway is converted back (same values), not refused."""
    import copy
    from mst_amd import _lib
    from test_encoder_gpu import build_model
    cfg = cases.CFG_DEFAULT
    base, _ = build_model(cfg)
    B, T = 2, 44100
    d = _stems(B, T, seed=4)
    feats = torch.randn(B, 64, generator=torch.Generator().manual_seed(5)).cuda()
    outs = []
    for how in ("reference", "cm32"):
        m = copy.deepcopy(base).train()
        m.train_backend, m.train_precision = "hip-strict", "fp32"
        assert m._train_encoder().train_layout() == _lib.LOGMEL_REF
        plan = m.audio_encoder.mel_preprocessor.plan(0)
        lm, _ = plan.forward_stems(d, True, False, _lib.LOGMEL_REF if how == "reference" else _lib.LOGMEL_CM32)
        torch.manual_seed(3)
        emb = m.forward_from_logmel(lm, feats)
        emb.sum().backward()
        outs.append((emb.detach().clone(), m.film_encoder.film_head.weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    enc = copy.deepcopy(base).train()._train_encoder()
    lm16, _ = base.audio_encoder.mel_preprocessor.plan(0).forward_stems(d, True, False, _lib.LOGMEL_CM16, want_absmax=True)
    with pytest.raises(_lib.MstError, match="layout"):
        enc.forward_train(lm16, feats=feats)   # straight at the C entry: the fp32 mode refuses the float16 planes


@pytest.mark.parametrize("precision", ["f16", "f16x3"])
def test_conv2_weight_gradient_from_the_float16_pool1_planes(precision):
    """`mst_encoder_train_conv2_wgrad(pool1 = NULL)` (float16 training modes): conv2's weight gradient reads the float16
    pool1 planes the pooling epilogue left in the workspace instead of rounding the fp32 tensor again -- the same bits, so the
    gradient is bit-identical -- and a forward pass that is not asked for the fp32 pool1 does not produce it."""
    from test_encoder_gpu import build_model
    cfg = cases.CFG_DEFAULT
    m, _ = build_model(cfg)
    enc = m.train()._train_encoder()
    enc.set_train_precision(precision)
    flat, _ = m._trunk_flat()
    enc.update_trunk_params(*flat)
    B, T = 3, 44100
    plan = m.audio_encoder.mel_preprocessor.plan(0)
    lm, _ = plan.forward_stems(_stems(B, T, seed=6), True, False)
    g = torch.Generator().manual_seed(8)
    film = (torch.randn(B, enc.n_sub * 192, generator=g) * 0.3 + 1.0).cuda()
    Fr = lm.shape[-1]
    grads = []
    for want in (True, False):
        enc._ws_train = None
        _, t = enc.forward_train(lm, film=film, head=False, drop1_p=0.3, drop1_seed=5, want_pool1=want)
        assert (t["pool1"] is not None) == want
        dp = torch.randn(t["pool_in"].shape, generator=g).cuda() if not grads else dp   # noqa: F821
        dfilm = torch.zeros(B, enc.n_sub * 192, device="cuda")
        enc.backward_apply(2, dp, dfilm, B, Fr)
        grads.append(enc.conv2_wgrad(t["pool1"], B, Fr))
    torch.cuda.synchronize()
    assert torch.isfinite(grads[0]).all() and grads[0].abs().max() > 0
    assert torch.equal(grads[0], grads[1])
    enc.set_train_precision("fp32")
    enc._ws_train = None
    with pytest.raises(Exception, match="float16"):
        enc.forward_train(lm, film=film, head=False, want_pool1=False)


@pytest.mark.parametrize("precision", ["fp32", "f16x3-all", "f16"])
def test_eval_encoder_is_refreshed_on_the_device_after_parameter_updates(precision):
    """src/train.py:388-427 (validate_epoch) follows :292-296 (optimizer steps): the eval encoder must see the new parameters.
    `hip_encoder()` keeps its handle and rebuilds every table ON THE DEVICE (`mst_encoder_update_params`: MFMA fragment swizzles,
    eval BatchNorm fold, float16 fragments + pre-scales, transposes) -- no parameter crosses to the host (checked with torch's
    sync debug mode) -- and the result equals an encoder built from scratch from the same state_dict."""
    cfg = cases.CFG_DEFAULT
    m = _build(cfg, precision)
    d = _stems(2, 44100, seed=9)
    feats = torch.randn(2, 64, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        e_before = m(d, feats)
        first = m._hip
        g = torch.Generator(device="cuda").manual_seed(5)
        for p in m.parameters():                                  # "optimizer steps"
            p.add_(torch.randn(p.shape, generator=g, device="cuda") * (0.02 * p.abs().mean()))
        for name, b in m.named_buffers():                         # running statistics move too
            if "running_mean" in name:
                b.add_(0.05)
            elif "running_var" in name:
                b.mul_(1.1)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            enc = m.hip_encoder()                                 # the refresh: device-side only
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert enc is first, "the handle is kept, its tables are rebuilt"
        e_after = m(d, feats)
        fresh = _build(cfg, precision)                            # another module, its encoder built from scratch through the host path
        fresh.load_state_dict(m.state_dict())
        e_fresh = fresh(d, feats)
    assert fresh._hip is not first
    assert (e_after - e_before).abs().max() > 1e-4, "the parameters did change"
    if precision == "fp32":
        assert torch.equal(e_after, e_fresh)
    else:   # the L1 norms behind the range scales are summed in another order on the device: at most a last-bit difference
        assert (e_after - e_fresh).abs().max().item() <= 2e-6 * e_fresh.abs().max().item()


def test_eval_encoder_sees_running_statistics_updated_without_an_optimizer_step():
    """BatchNorm recalibration / a GradScaler-skipped step right before validation: train-mode forwards move the running
    statistics (written by `mst_encoder_train_update_running_stats` through raw pointers, which bumps no tensor version) while no
    parameter changes.  The next eval forward must fold the NEW statistics -- equal to the PyTorch-ROCm backend's eval forward of the
    same module, and different from the embeddings before the recalibration."""
    cfg = cases.CFG_DEFAULT
    m = _build(cfg, "fp32")
    d = _stems(3, 44100, seed=11)
    feats = torch.randn(3, 64, generator=torch.Generator().manual_seed(2)).cuda()
    with torch.no_grad():
        e_before = m(d, feats).clone()
    rm_before = m.audio_encoder.subnet_cnns[3].bn1.running_mean.clone()
    m.train()
    for _ in range(3):               # forward only: no backward, no optimizer step
        m(d, feats)
    m.eval()
    assert (m.audio_encoder.subnet_cnns[3].bn1.running_mean - rm_before).abs().max() > 1e-3, "the statistics did move"
    with torch.no_grad():
        e_after = m(d, feats)
        m.encoder_backend = "torch"
        e_torch = m(d, feats)
        m.encoder_backend = "hip"
    assert (e_after - e_before).abs().max() > 1e-4 * e_before.abs().max(), "the eval encoder still folds the old statistics"
    assert (e_after - e_torch).abs().max().item() <= 1e-4 * e_torch.abs().max().item()


def test_plain_float16_consumers_get_the_high_parts_alone():
    """`forward_stems(..., LOGMEL_CM16, want_lo=False)`: stage A writes the float16 high parts only (the plain-float16 eval mode and
    the f16 training mode read nothing else) -- same bits as with the low parts, a quarter of the log-mel bytes less to write and to
    keep alive; such a LogMel cannot be turned back into the fp32 tensor and says so; the module asks for it by itself."""
    from mst_amd import _lib
    plan = _plan()
    d = _stems(2, 44100, seed=8)
    full, _ = plan.forward_stems(d, True, False, _lib.LOGMEL_CM16, want_absmax=True)
    hi, _ = plan.forward_stems(d, True, False, _lib.LOGMEL_CM16, want_absmax=True, want_lo=False)
    assert hi.lo is None and torch.equal(hi.data, full.data) and torch.equal(hi.absmax, full.absmax)
    with pytest.raises(_lib.MstError, match="HIGH parts"):
        hi.to_reference()
    m = _build(cases.CFG_DEFAULT, "f16")
    feats = torch.randn(2, 64, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        e_hi = m.hip_encoder().forward(hi, feats)
        e_full = m.hip_encoder().forward(full, feats)
    assert torch.equal(e_hi, e_full)
    with pytest.raises(_lib.MstError, match="low parts"):
        _build(cases.CFG_DEFAULT, "f16x3-all").hip_encoder().forward(hi, feats)
    seen = []
    orig = plan.__class__.forward_stems

    def spy(self, *a, **k):
        seen.append(k.get("want_lo", True))
        return orig(self, *a, **k)
    plan.__class__.forward_stems = spy
    try:
        with torch.no_grad():
            m(d, feats)
        mt = _build(cases.CFG_DEFAULT).train()
        mt.train_backend, mt.train_precision = "hip-strict", "f16"
        mt(d, feats).sum().backward()
        mt.train_precision = "f16x3"
        mt(d, feats).sum().backward()
    finally:
        plan.__class__.forward_stems = orig
    assert seen == [False, False, True], seen
