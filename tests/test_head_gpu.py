"""GPU parity of the hand-written training forward / backward of the attention-pooling head and the FiLM MLP
(csrc/head.hip: `mst_head_forward_train`, `mst_head_backward`, `mst_film_forward_train`, `mst_film_backward`).

Oracle: the SAME arithmetic written with torch modules of the reference's structure -- F.dropout -> AttentionPooling
(src/model.py:118, :187-211) and MixingFeatureEncoder's MLP (src/model.py:385-464) -- evaluated in FLOAT64 with autograd, with
the Dropout masks the kernels derive (made explicit through `mst_dropout_mask`).  Tolerance 2e-5 of each tensor's maximum (fp32
sums of up to 6192 products against float64).  The whole training step, these networks included, is pinned to the reference's
own training arithmetic by tests/test_encoder_gpu.py::test_hip_training_step_matches_the_reference_training_fixture."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mask(p, seed, shape):
    from mst_amd import _lib
    n = int(np.prod(shape))
    keep = torch.empty(n, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().mst_dropout_mask(float(p), int(seed), n, _lib.dptr(keep), _lib.stream_ptr(keep.device)), "mst_dropout_mask")
    return keep.view(*shape).double() / (1.0 - p)


def _close(a, ref, tol, name):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    scale = ref.abs().max().item()
    err = (a - ref).abs().max().item()
    assert err <= tol * max(scale, 1e-30), f"{name}: max |d| {err:.3e} vs scale {scale:.3e}"


def _seeds(n):
    """The seeds the autograd Functions will draw next (they take them from torch's CPU generator)."""
    st = torch.get_rng_state()
    out = [int(torch.randint(0, 2 ** 62, (1,)).item()) for _ in range(n)]
    torch.set_rng_state(st)
    return out


@pytest.mark.parametrize("B,Cc,T,p_in,p_out", [(5, 1408, 86, 0.0, 0.0), (5, 1408, 86, 0.3, 0.3), (3, 192, 37, 0.3, 0.0), (2, 3072, 20, 0.0, 0.3)])
def test_attention_pooling_head_forward_backward(B, Cc, T, p_in, p_out):
    from mst_amd.model import AttentionPooling, _HipHead
    torch.manual_seed(B * 1000 + T)
    head = AttentionPooling(Cc, 256, 768).cuda()
    with torch.no_grad():
        head.attention[2].weight.mul_(3.0)     # a peaky softmax: the scores matter
    x = (torch.randn(B, Cc, T, device="cuda") * 0.7).relu_()     # pool_in is a ReLU / max-pool output
    R = torch.randn(B, 768, device="cuda")
    ps = [head.attention[0].weight, head.attention[0].bias, head.attention[2].weight, head.attention[2].bias,
          head.projection[0].weight, head.projection[0].bias]
    torch.manual_seed(99)
    s_in, s_out = (_seeds(2) + [0])[:2] if p_in > 0 and p_out > 0 else ((_seeds(1)[0], 0) if p_in > 0 else (0, _seeds(1)[0] if p_out > 0 else 0))
    xg = x.clone().requires_grad_(True)
    emb = _HipHead.apply(xg, p_in, p_out, *ps)
    (emb * R).sum().backward()
    got = [xg.grad.clone()] + [q.grad.clone() for q in ps]
    for q in ps:
        q.grad = None
    # float64 oracle with explicit masks
    h64 = AttentionPooling(Cc, 256, 768).cuda().double()
    h64.load_state_dict({k: v.double() for k, v in head.state_dict().items()})
    m_in = _mask(p_in, s_in, (B, Cc, T)) if p_in > 0 else 1.0
    m_out = _mask(p_out, s_out, (B, 768)) if p_out > 0 else 1.0
    x64 = x.double().requires_grad_(True)
    xt = (x64 * m_in).transpose(1, 2)
    w = torch.softmax(h64.attention(xt), dim=1)
    ref = torch.relu(h64.projection[0]((xt * w).sum(dim=1))) * m_out
    (ref * R.double()).sum().backward()
    want = [x64.grad] + [q.grad for q in (h64.attention[0].weight, h64.attention[0].bias, h64.attention[2].weight, h64.attention[2].bias,
                                          h64.projection[0].weight, h64.projection[0].bias)]
    _close(emb, ref, 2e-5, "embedding")
    if p_out > 0:
        keep = (m_out > 0).float().mean().item()
        assert abs(keep - (1 - p_out)) < 4 * (p_out * (1 - p_out) / (B * 768)) ** 0.5 + 1e-3, keep
    names = ["d pool_in", "attention.0.weight", "attention.0.bias", "attention.2.weight", "attention.2.bias", "projection.0.weight",
             "projection.0.bias"]
    for g, r, n in zip(got, want, names):
        if n == "attention.2.bias":     # softmax is shift-invariant: the gradient is 0 up to rounding
            assert g.abs().max().item() <= 1e-4 * max(1.0, want[3].abs().max().item()), g
            continue
        _close(g, r, 5e-5 if n != "d pool_in" else 2e-5, n)
    # determinism: the same seed gives the same bits
    torch.manual_seed(99)
    xg2 = x.clone().requires_grad_(True)
    emb2 = _HipHead.apply(xg2, p_in, p_out, *ps)
    (emb2 * R).sum().backward()
    assert torch.equal(emb, emb2) and torch.equal(xg2.grad, got[0]) and torch.equal(ps[0].grad, got[1])


@pytest.mark.parametrize("B,Fd,n_sub,p", [(6, 64, 11, 0.0), (6, 64, 11, 0.2), (3, 180, 24, 0.2)])
def test_film_mlp_forward_backward(B, Fd, n_sub, p):
    from mst_amd.model import MixingFeatureEncoder, _HipFilmMLP
    torch.manual_seed(7 + B)
    fe = MixingFeatureEncoder(Fd, n_sub).cuda()
    f = torch.randn(B, Fd, device="cuda") * 2.0
    R = torch.randn(B, n_sub * 192, device="cuda")
    mlp = fe.feature_mlp
    ps = [mlp[0].weight, mlp[0].bias, mlp[3].weight, mlp[3].bias, fe.film_head.weight, fe.film_head.bias]
    torch.manual_seed(5)
    seed = _seeds(1)[0] if p > 0 else 0
    film = _HipFilmMLP.apply(f, p, *ps)
    (film * R).sum().backward()
    got = [q.grad.clone() for q in ps]
    fe64 = MixingFeatureEncoder(Fd, n_sub).cuda().double()
    fe64.load_state_dict({k: v.double() for k, v in fe.state_dict().items()})
    m = _mask(p, seed, (B, 256)) if p > 0 else 1.0
    h1 = torch.relu(fe64.feature_mlp[0](f.double())) * m
    ref = fe64.film_head(torch.relu(fe64.feature_mlp[3](h1)))
    (ref * R.double()).sum().backward()
    want = [fe64.feature_mlp[0].weight.grad, fe64.feature_mlp[0].bias.grad, fe64.feature_mlp[3].weight.grad, fe64.feature_mlp[3].bias.grad,
            fe64.film_head.weight.grad, fe64.film_head.bias.grad]
    _close(film, ref, 1e-5, "film")
    for g, r, n in zip(got, want, ["mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias", "film_head.weight", "film_head.bias"]):
        _close(g, r, 2e-5, n)


def test_training_step_uses_the_hand_written_small_networks_and_matches_the_module_path():
    """model.train() step: with `small_nets_backend = "hip"` (default) the head and the FiLM MLP run in libmst.so; the step equals the
    step with the nn.Modules + autograd (Dropout off so that no mask stream has to match) to fp32 rounding."""
    import cases
    from oracle import mel as omel
    from test_encoder_gpu import build_model
    cfg = cases.CFG_DEFAULT
    B, T = 4, 44100
    x = torch.stack([cases.synth_clip(c, T) for c in range(B)], 0).cuda()
    lm = None
    feats = torch.randn(B, 64, generator=torch.Generator().manual_seed(2)).cuda()
    R = torch.randn(B, cfg["embed_dim"], generator=torch.Generator().manual_seed(3)).cuda()
    res = {}
    for backend in ("hip", "torch"):
        m, _ = build_model(cfg)      # the same seeded parameters both times
        for q in m.modules():
            if isinstance(q, torch.nn.Dropout):
                q.p = 0.0
        m.train()
        m.train_backend, m.small_nets_backend = "hip-strict", backend
        if lm is None:
            with torch.no_grad():
                lm = m.audio_encoder.mel_preprocessor(omel.tensor_to_stems_dict(x))
        emb = m.forward_from_logmel(lm, feats)
        (emb * R).sum().backward()
        res[backend] = (emb.detach(), {n: p.grad.detach() for n, p in m.named_parameters()})
    _close(res["hip"][0], res["torch"][0], 2e-5, "embedding")
    for n, g in res["torch"][1].items():
        den = g.abs().max().item()
        if den > 1e-12 and not n.endswith(("conv1.bias", "conv2.bias", "attention.2.bias")):
            _close(res["hip"][1][n], g, 3e-4, n)


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_training_step_is_a_descent_direction(precision):
    """End-to-end gradient check of the whole product step (stage A -> trunk -> head, FiLM MLP -> InfoNCE, everything in libmst.so,
    batch-statistics BatchNorm, Dropout off so that the function is the same at both points): moving every parameter by
    -eta * gradient changes the loss by -eta * |g|^2 to first order.  eta is chosen for a predicted decrease of 1 % of the loss; the
    measured decrease must be within 15 % of it (curvature, max-pool / ReLU kinks; measured 0.5 %)."""
    import cases
    from mst_amd.loss import InfoNCELoss
    from oracle import mel as omel
    from test_encoder_gpu import build_model
    cfg = cases.CFG_DEFAULT
    m, _ = build_model(cfg)
    for q in m.modules():
        if isinstance(q, torch.nn.Dropout):
            q.p = 0.0
    m.train()
    m.train_backend, m.train_precision = "hip-strict", precision
    B, T = 8, 44100
    x = torch.stack([cases.synth_clip(c % 4, T) * (1.0 + 0.1 * c) for c in range(B)], 0).cuda()
    stems = omel.tensor_to_stems_dict(x)
    feats = (torch.randn(B, 64, generator=torch.Generator().manual_seed(4)) * 1.5).cuda()
    labels = (torch.arange(B) // 2).cuda()
    crit = InfoNCELoss(0.1)
    l0 = crit(m(stems, feats), labels)
    l0.backward()
    g2 = sum((p.grad.double() ** 2).sum().item() for p in m.parameters())
    eta = 0.01 * l0.item() / g2
    with torch.no_grad():
        for p in m.parameters():
            p.add_(p.grad, alpha=-eta)
        for q in m.modules():     # (the running statistics moved; they do not enter a training-mode forward)
            pass
    l1 = crit(m(stems, feats), labels)
    pred, got = eta * g2, l0.item() - l1.item()
    print(f"descent check [{precision}]: loss {l0.item():.5f} -> {l1.item():.5f}, predicted decrease {pred:.5f}, measured {got:.5f}")
    assert 0.85 * pred <= got <= 1.15 * pred, (l0.item(), l1.item(), pred, got)   # measured: 1.005
