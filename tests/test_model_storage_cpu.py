"""CPU: the stacked-storage views behind the training step (MixingStyleEncoder._trunk_flat / _bn_flat) keep the reference's module
surface intact -- state_dict keys, shapes and values (tests/test_oracle_golden.py pins the key list to the reference's), load /
deepcopy / save round trips -- while the eight trunk parameter families live in ONE storage (a training pass snapshots them with
one copy) and the BatchNorm buffers in three stacked tensors per layer (one running-statistics kernel per layer)."""
import copy
import io

import torch


def _model():
    from mst_amd.model import MixingStyleEncoder
    torch.manual_seed(0)
    return MixingStyleEncoder(feature_dim=64)


def test_trunk_parameters_become_views_of_one_storage_without_changing_the_module_surface():
    m = _model()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    flat, params = m._trunk_flat()
    assert len(flat) == 8 and len(params) == 8 * m.audio_encoder.n_subbands
    base = flat[0]._base
    assert base is not None and all(f._base is base for f in flat) and all(f.is_contiguous() for f in flat)
    assert base.numel() == sum(f.numel() for f in flat)
    after = m.state_dict()
    assert list(after.keys()) == list(before.keys()) and [n for n, _ in m.named_parameters()] == names
    assert all(torch.equal(after[k], before[k]) and after[k].shape == before[k].shape for k in before)
    # a parameter IS a view: writing the stacked tensor is writing the module
    with torch.no_grad():
        flat[0][3].add_(1.0)
    assert torch.equal(m.audio_encoder.subnet_cnns[3].conv1.weight, before["audio_encoder.subnet_cnns.3.conv1.weight"] + 1.0)
    flat2, _ = m._trunk_flat()
    assert all(a is b for a, b in zip(flat, flat2)), "a second call finds the storage in place"
    # optimizers see ordinary leaf parameters
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert torch.isfinite(flat[4]).all() and not torch.equal(flat[4][0], before["audio_encoder.subnet_cnns.0.conv2.weight"])


def test_batchnorm_buffers_as_stacked_views_round_trip():
    m = _model()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    rm, rv, nb = m._bn_flat("bn2")
    assert rm.shape == (m.audio_encoder.n_subbands, 64) and nb.dtype == torch.int64
    rm.add_(0.5), nb.add_(2)
    sd = m.state_dict()
    k = "audio_encoder.subnet_cnns.7.bn2.running_mean"
    assert torch.equal(sd[k], before[k] + 0.5) and sd["audio_encoder.subnet_cnns.7.bn2.num_batches_tracked"].item() == 2
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    m2 = _model()
    m2.load_state_dict(torch.load(buf))
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), sd.values()))
    m3 = copy.deepcopy(m)
    assert all(torch.equal(a, b) for a, b in zip(m3.state_dict().values(), sd.values()))
    m3._bn_flat("bn2")[0].zero_()                      # the copy has its own storage
    assert torch.equal(m.state_dict()[k], before[k] + 0.5)
    m.load_state_dict(before)                          # loading writes through the views
    assert torch.equal(rm[7], before[k])
    # re-materialised buffers (e.g. after .to(dtype)) are re-stacked on the next call
    m.double()
    rm64, _, _ = m._bn_flat("bn2")
    assert rm64.dtype == torch.float64 and rm64 is not rm and torch.equal(rm64[7].float(), before[k])
