"""The reference's only tests for this path are the shape checks of inference/test_model.py:14-178 (no numeric
expectations).  Same checks, same sizes, against the MI355X modules (reads like the reference's file)."""
import pytest
import torch

import cases  # noqa: F401

pytestmark = pytest.mark.gpu


def test_attention_pooling():                      # inference/test_model.py:14-41
    from mst_amd.model import AttentionPooling
    features = torch.randn(4, 512, 25).cuda()
    out = AttentionPooling(input_dim=512, hidden_dim=128, output_dim=768).cuda()(features)
    assert out.shape == (4, 768)


def test_mel_preprocessor():                       # inference/test_model.py:44-89
    from mst_amd.model import MelSpectrogramPreprocessor
    T = int(44100 * 10.0)
    stems = {k: torch.randn(2, 2, T).cuda() for k in ("vocals", "bass", "drums", "other")}
    mel = MelSpectrogramPreprocessor(sample_rate=44100, n_fft=1024, hop_length=256, n_mels=128)(stems)
    assert mel.shape[0] == 2 and mel.shape[1] == 8 and mel.shape[2] == 128 and mel.shape[3] == T // 256 + 1


def test_full_model_with_raw_audio():              # inference/test_model.py:92-147 (feature_dim=100, eval forward)
    from mst_amd.model import MixingStyleEncoder
    T = int(44100 * 10.0)
    stems = {k: torch.randn(2, 2, T).cuda() for k in ("vocals", "bass", "drums", "other")}
    model = MixingStyleEncoder(sample_rate=44100, n_fft=1024, hop_length=256, n_mels=128, split_size=20, overlap=10,
                               channels=8, embed_dim=768, feature_dim=100).cuda().eval()
    with torch.no_grad():
        emb = model(stems, torch.randn(2, 100).cuda())
    assert emb.shape == (2, 768) and torch.isfinite(emb).all()
    assert sum(p.numel() for p in model.parameters()) == 3313313 + (100 - 64) * 256   # SURVEY A.2 at feature_dim=64


def test_attention_weights():                      # inference/test_model.py:150-178
    from mst_amd.model import AttentionPooling
    pool = AttentionPooling(input_dim=256, hidden_dim=128, output_dim=768).cuda()
    features = torch.randn(3, 256, 30).cuda()
    w = torch.softmax(pool.attention(features.transpose(1, 2)), dim=1)
    assert torch.allclose(w.sum(dim=1), torch.ones(3, 1).cuda())
