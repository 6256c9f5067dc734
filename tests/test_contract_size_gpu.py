"""GPU: the contract workload checked at ITS OWN size -- 24 triplets = 72 clips of 10 s (441 000 samples), the batch `bench.py`
times (reference: src/model.py:508-542 at src/params.py:43's batch size x 3 clips per triplet).

The 72-clip launches take index paths no small parity case executes: `conv1_resident_kernel` deals nsub x sets_per_band sets
over 256 persistent workgroups (the sub-band changes INSIDE a workgroup's run, clips straddle workgroup boundaries, the last
set of a band is partial), conv2 deals 4 257 sets round-robin, stage A runs 32 x 72 workgroups in nine rounds.  So:
  * every clip's embedding from the B = 72 call must equal, BIT FOR BIT, the embedding of the same clip run alone (B = 1);
  * the first clip, the clips on either side of every band change, a middle one and the last are compared with the CPU oracle
    (`oracle.encoder.encoder_forward` on the same waveform) element-wise at 1e-4;
for the exact-fp32 kernels and for the split-precision mode (`f16x3-all`)."""
import numpy as np
import pytest
import torch

import parity
from oracle import encoder as oenc
from oracle import features as ofeat

pytestmark = pytest.mark.gpu
B, T = 72, 441000


@pytest.fixture(scope="module")
def contract():
    """bench.py's model (seed 42, FiLM gammas ~ 1) and its batch (`synth_batch`, generated on the device)."""
    from mst_amd.mixing_utils import STEMS, deferred_features
    from mst_amd.model import MixingStyleEncoder
    from mst_amd.synth import synth_batch
    torch.manual_seed(42)
    model = MixingStyleEncoder(44100, 1024, 256, 128, 20, 10, 8, 768, feature_dim=64)
    with torch.no_grad():
        b = model.film_encoder.film_head.bias
        for i in range(model.audio_encoder.n_subbands):
            b[i * 192:i * 192 + 32] += 1.0
            b[i * 192 + 64:i * 192 + 128] += 1.0
    model = model.cuda().eval()
    x = synth_batch(B, T, device="cuda")
    stems = {s: x[:, 2 * i:2 * i + 2] for i, s in enumerate(STEMS)}
    deferred = torch.stack([deferred_features(64)] * B).cuda()
    return model, x, stems, deferred


def _oracle_clips():
    # conv1's sets are band-major, clip-major inside a band: the band changes between clip 71 and clip 0; workgroup boundaries
    # fall inside arbitrary clips (42 570 sets over 256 workgroups) -- first, last, neighbours of the wrap, and a middle clip
    return [0, 1, 35, 70, 71]


@pytest.mark.parametrize("precision", ["fp32", "f16x3-all"])
def test_contract_batch_is_batch_independent_and_matches_the_oracle(contract, precision):
    model, x, stems, deferred = contract
    model.conv1_precision = precision
    try:
        with torch.no_grad():
            emb = model(stems, deferred)                                   # the trainer's call: deferred rows + stems, B = 72
            assert tuple(emb.shape) == (B, 768) and bool(torch.isfinite(emb).all())
            bad = []
            for c in range(B):
                one = {k: v[c:c + 1] for k, v in stems.items()}
                e1 = model(one, deferred[c:c + 1])
                if not torch.equal(e1[0], emb[c]):
                    bad.append((c, float((e1[0] - emb[c]).abs().max())))
        assert not bad, f"{len(bad)} of {B} clips differ between the B = 72 launch and B = 1: {bad[:8]}"
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        clips = _oracle_clips()
        xs = x[clips].cpu()
        ref = oenc.encoder_forward(sd, xs, ofeat.extract_all_features(xs))
        got = emb[clips].cpu()
        for i, c in enumerate(clips):
            r, g = ref[i].double().numpy(), got[i].double().numpy()
            rel = np.abs(g - r) / np.maximum(np.abs(r), 1e-2 * np.abs(r).max())
            parity.record(f"contract size B=72 clip {c} [{precision}] embedding [element-wise]", g, r)
            assert rel.max() <= 1e-4, f"clip {c}: worst element {rel.max():.2e} beyond 1e-4 ({int((rel > 1e-4).sum())} elements)"
    finally:
        model.conv1_precision = "fp32"


def test_contract_batch_stage_taps_match_single_clip_runs(contract):
    """The intermediate tensors of the B = 72 launch (pool1 after conv1, pool_in after conv2) against B = 1 runs of the clips on
    either side of a band change and of a workgroup boundary: a wrong tile index that happened to cancel in the embedding would
    show here."""
    model, x, stems, _ = contract
    from mst_amd.mixing_utils import MixingFeatureExtractor
    fe = MixingFeatureExtractor()
    with torch.no_grad():
        feats, lm = fe.features_and_logmel(stems)
        _, taps = model.hip_encoder().forward(lm, feats, taps=True)
        for c in (0, 17, 71):
            f1, l1 = fe.features_and_logmel({k: v[c:c + 1] for k, v in stems.items()})
            assert torch.equal(f1[0], feats[c]) and torch.equal(l1[0], lm[c])
            _, t1 = model.hip_encoder().forward(l1, f1, taps=True)
            for k in ("film", "pool1", "pool_in"):
                assert torch.equal(t1[k][0], taps[k][c]), (c, k)
