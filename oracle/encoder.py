"""Oracle: FiLM MLP + band-split CNN + attention pooling, eval mode (CPU, fp32).

Functional restatement of reference src/model.py:70-542 driven directly by a
reference-format `state_dict` (keys as produced by MixingStyleEncoder.state_dict()).
Dropout layers are identity (eval); BatchNorm uses running statistics.
"""
import torch
import torch.nn.functional as F

from . import mel as omel


def n_subbands(n_mels: int, split_size: int, overlap: int) -> int:
    """model.py:257-261"""
    n, i = 0, 0
    while overlap * i <= n_mels - split_size:
        n += 1
        i += 1
    return n


def film_params(sd, feats: torch.Tensor) -> torch.Tensor:
    """model.py:410-464 -> flat (B, n_sub*192); per band [g1(32) b1(32) g2(64) b2(64)]."""
    p = "film_encoder."
    h = F.relu(F.linear(feats, sd[p + "feature_mlp.0.weight"], sd[p + "feature_mlp.0.bias"]))
    h = F.relu(F.linear(h, sd[p + "feature_mlp.3.weight"], sd[p + "feature_mlp.3.bias"]))
    return F.linear(h, sd[p + "film_head.weight"], sd[p + "film_head.bias"])


def round_f16_ideal(t: torch.Tensor) -> torch.Tensor:
    """nearest-even rounding to 11 significant bits (float16's precision) with the exponent range of the input dtype"""
    m, e = torch.frexp(t)                      # |m| in [0.5, 1)
    return torch.ldexp(torch.round(m * 2048.0) / 2048.0, e)


def _f16_operand(t: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """value after a round trip through float16.  The kernels multiply every operand by an exact power of two before the
    conversion (weights per output channel, activations per band) so that neither float16's ceiling nor its subnormals
    are reached: the round trip is a rounding to 11 significant bits, whatever the magnitude (`scale` is kept for callers
    of the first version, which pre-scaled by a fixed 2^10)."""
    return round_f16_ideal(t)


def subband_cnn(sd, i: int, x: torch.Tensor, film: torch.Tensor, split_size: int,
                taps=None, f16_operands=False, bn_training=False, f16_outputs=False) -> torch.Tensor:
    """model.py:127-157 for sub-band i.  x (B, 8, split, F) -> (B, 64, H', W').
    f16_operands: the arithmetic of the opt-in "f16" precision mode (and of the reference under `--use_amp` autocast,
    src/train.py:251-253, for the convolutions): conv inputs and weights rounded to float16, products and sums in fp32."""
    p = f"audio_encoder.subnet_cnns.{i}."
    g1, b1, g2, b2 = torch.split(film[:, i * 192:(i + 1) * 192], [32, 32, 64, 64], dim=1)
    sub = max(1, split_size // 10)

    def block(x, conv, bn, g, b, pool):
        w = sd[p + conv + ".weight"]
        if f16_operands:
            x, w = _f16_operand(x), _f16_operand(w, 1024.0)
        x = F.conv2d(x, w, sd[p + conv + ".bias"], padding=3)
        if f16_outputs:   # f16 TRAINING mode: the convolution output is stored as float16 (as under torch.autocast), statistics of that
            x = round_f16_ideal(x)
        if taps is not None and x.requires_grad:   # gradient checks: keep the convolution output and its gradient
            x.retain_grad()
            taps[f"{conv}_out_{i}"] = x
        if bn_training:   # nn.BatchNorm2d under model.train(): statistics of this batch, biased variance
            if taps is not None:
                taps[f"{bn}_{i}"] = (x.mean(dim=(0, 2, 3)), x.var(dim=(0, 2, 3), unbiased=False))
            x = F.batch_norm(x, None, None, sd[p + bn + ".weight"], sd[p + bn + ".bias"], training=True, eps=1e-5)
        else:
            x = F.batch_norm(x, sd[p + bn + ".running_mean"], sd[p + bn + ".running_var"],
                             sd[p + bn + ".weight"], sd[p + bn + ".bias"], training=False, eps=1e-5)
        x = g[:, :, None, None] * x + b[:, :, None, None]
        return F.max_pool2d(F.relu(x), pool)

    y1 = block(x, "conv1", "bn1", g1, b1, (sub, 5))
    y2 = block(y1, "conv2", "bn2", g2, b2, (4, 4))
    if taps is not None:
        taps[f"pool1_{i}"] = y1
        taps[f"pool2_{i}"] = y2
    return y2


def attention_pool(sd, x: torch.Tensor) -> torch.Tensor:
    """model.py:187-211.  x (B, C, T') -> (B, embed)."""
    p = "audio_encoder.attention_pooling."
    xt = x.transpose(1, 2)
    s = F.linear(torch.tanh(F.linear(xt, sd[p + "attention.0.weight"], sd[p + "attention.0.bias"])),
                 sd[p + "attention.2.weight"], sd[p + "attention.2.bias"])
    w = torch.softmax(s, dim=1)
    pooled = (xt * w).sum(dim=1)
    return F.relu(F.linear(pooled, sd[p + "projection.0.weight"], sd[p + "projection.0.bias"]))


def encoder_from_logmel(sd, lm: torch.Tensor, feats: torch.Tensor, split_size=20, overlap=10,
                        taps=None, f16_operands=False, bn_training=False, f16_outputs=False) -> torch.Tensor:
    """log-mel (B, 8, M, F), features (B, Fd) -> embeddings (B, E).  model.py:290-382,508-542."""
    film = film_params(sd, feats)
    nsub = n_subbands(lm.shape[2], split_size, overlap)
    outs = [subband_cnn(sd, i, lm[:, :, i * overlap:i * overlap + split_size, :], film, split_size, taps, f16_operands, bn_training,
                        f16_outputs)
            for i in range(nsub)]
    cat = torch.cat(outs, dim=1)  # (B, nsub*64, H', W')
    flat = cat.reshape(cat.shape[0], cat.shape[1] * cat.shape[2], cat.shape[3])
    if taps is not None:
        taps["film"] = film
        taps["pool_in"] = flat
    return attention_pool(sd, flat)


def encoder_forward(sd, stems: torch.Tensor, feats: torch.Tensor, sample_rate=44100, n_fft=1024,
                    hop=256, n_mels=128, split_size=20, overlap=10, taps=None) -> torch.Tensor:
    """stems (B, 8, T) -> embeddings; mel front end + encoder_from_logmel."""
    lm = omel.logmel(stems, sample_rate, n_fft, hop, n_mels)
    if taps is not None:
        taps["logmel"] = lm
    return encoder_from_logmel(sd, lm, feats, split_size, overlap, taps)
