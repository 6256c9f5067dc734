"""CPU: libmst.so builds, loads, and exports every symbol include/mst.h declares (no compute calls)."""
import os
import re

import pytest

import cases  # noqa: F401  (sys.path setup)
from mst_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mst.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mst_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(built):
    decl = declared_symbols()
    assert decl, "no declarations parsed"
    assert set(decl) == set(_lib.SYMBOLS), (set(decl) ^ set(_lib.SYMBOLS))
    for name in decl:
        assert hasattr(built, name), f"libmst.so does not export {name}"


def test_version_and_error_string(built):
    assert built.mst_version() == 1
    assert isinstance(built.mst_last_error(), bytes)


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(_lib.EncoderConfig) == 9 * 4
    assert C.sizeof(_lib.EncoderWeights) == 24 * C.sizeof(C.c_void_p)
    assert C.sizeof(_lib.AugStem) == 16 + 18 * 8 + 8
    assert C.sizeof(_lib.AugClip) == 4 * C.sizeof(_lib.AugStem) + 8


def test_product_refuses_cpu_tensors():
    """The product path must fail loudly without the GPU (no CPU fallback)."""
    import torch
    from mst_amd.mixing_utils import MixingFeatureExtractor
    fe = MixingFeatureExtractor()
    assert fe.get_feature_dim() == 64
    assert MixingFeatureExtractor(use_detailed_spectral=True, n_spectral_bins=32).get_feature_dim() == 180
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        fe.extract_all_features({k: torch.zeros(2, 4096) for k in ("vocals", "bass", "drums", "other")})


def test_host_filterbank_matches_golden():
    import numpy as np
    from mst_amd.mixing_utils import hann_window, melscale_fbanks_htk
    g = np.load(os.path.join(ROOT, "tests", "golden", "fbanks.npz"))
    for n_fft, n_mels in ((1024, 128), (1024, 256), (2048, 80)):
        assert np.array_equal(melscale_fbanks_htk(n_fft // 2 + 1, n_mels, 44100).numpy(), g[f"fb_{n_fft}_{n_mels}"])
        assert np.array_equal(hann_window(n_fft).numpy(), g[f"win_{n_fft}"])


def test_closed_form_butterworth_matches_scipy():
    """AudioAugmenter._butter4_low vs scipy.signal.butter(4, fc, 'low', output='sos'): same cascade to rounding."""
    import numpy as np
    from scipy.signal import butter, sosfilt
    from mst_amd.mixing_utils import AudioAugmenter
    aug = AudioAugmenter()
    x = np.random.default_rng(0).standard_normal(20000)
    for fc in (4000.0, 5123.456, 8000.0, 11999.9):
        mine, ref = aug._butter4_low(fc), butter(4, fc, btype="low", fs=44100, output="sos")
        tf = lambda sos: (np.convolve(sos[0, :3], sos[1, :3]), np.convolve(sos[0, 3:], sos[1, 3:]))
        (b1, a1), (b2, a2) = tf(mine), tf(ref)
        np.testing.assert_allclose(b1, b2, rtol=1e-12, atol=1e-16)
        np.testing.assert_allclose(a1, a2, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(sosfilt(mine, x), sosfilt(ref, x), rtol=0, atol=1e-12)
