"""ctypes binding of libmst.so (C ABI declared in include/mst.h) + in-tree build helper.

The product path fails loudly when the HIP library is missing or a call fails: there is no
CPU / PyTorch fallback for the kernels.
"""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MST_LIB", os.path.join(_HERE, "libmst.so"))  # MST_LIB: A/B-test another build
CSRC = os.path.join(_HERE, "csrc")

MST_OK = 0


class MstError(RuntimeError):
    pass


class EncoderConfig(C.Structure):
    _fields_ = [("n_mels", C.c_int32), ("split_size", C.c_int32), ("overlap", C.c_int32),
                ("n_subbands", C.c_int32), ("feature_dim", C.c_int32), ("embed_dim", C.c_int32),
                ("film_hidden", C.c_int32), ("attn_hidden", C.c_int32), ("bn_eps", C.c_float)]


_W_FIELDS = ["conv1_w", "conv1_b", "bn1_w", "bn1_b", "bn1_mean", "bn1_var",
             "conv2_w", "conv2_b", "bn2_w", "bn2_b", "bn2_mean", "bn2_var",
             "mlp0_w", "mlp0_b", "mlp3_w", "mlp3_b", "head_w", "head_b",
             "att0_w", "att0_b", "att2_w", "att2_b", "proj_w", "proj_b"]


class EncoderWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _W_FIELDS]


class EncoderTaps(C.Structure):
    _fields_ = [("film", C.c_void_p), ("pool1", C.c_void_p), ("pool_in", C.c_void_p), ("events", C.c_void_p * 6)]


# log-mel layouts (include/mst.h MST_LOGMEL_*)
LOGMEL_REF, LOGMEL_CM32, LOGMEL_CM16 = 0, 1, 2


class MelfeatIO(C.Structure):
    _fields_ = [("stems4", C.c_void_p * 4), ("clip_stride", C.c_longlong), ("pcm16", C.c_int32), ("layout", C.c_int32),
                ("logmel", C.c_void_p), ("logmel_lo", C.c_void_p), ("absmax", C.c_void_p), ("feats", C.c_void_p)]


class LogmelIn(C.Structure):
    _fields_ = [("layout", C.c_int32), ("pad_", C.c_int32), ("data", C.c_void_p), ("lo", C.c_void_p),
                ("absmax", C.c_void_p)]


class EncoderTrainTaps(C.Structure):
    _fields_ = [("film", C.c_void_p), ("pool1", C.c_void_p), ("pool_in", C.c_void_p), ("bn1", C.c_void_p),
                ("bn2", C.c_void_p), ("film_in", C.c_void_p), ("drop1_mask", C.c_void_p), ("drop1_scale", C.c_float),
                ("phase", C.c_int), ("count_scale", C.c_double), ("drop1_mask_out", C.c_void_p), ("drop1_seed", C.c_uint64),
                ("drop1_p", C.c_float)]


class AugStem(C.Structure):
    _fields_ = [("gain", C.c_float), ("tilt", C.c_int32), ("compress", C.c_int32), ("bw_sections", C.c_int32),
                ("tilt_sos", C.c_double * 6), ("bw_sos", C.c_double * 12), ("comp_threshold_db", C.c_float), ("comp_ratio", C.c_float)]


class AugClip(C.Structure):
    _fields_ = [("stem", AugStem * 4), ("reverb", C.c_int32), ("pad_", C.c_int32)]


# every symbol include/mst.h declares: (restype, argtypes)
class HeadDims(C.Structure):
    _fields_ = [("channels", C.c_int32), ("frames", C.c_int32), ("attn_hidden", C.c_int32), ("embed_dim", C.c_int32)]


class HeadPtrs(C.Structure):   # mst_head_weights / mst_head_grads
    _fields_ = [(k, C.c_void_p) for k in ("att0_w", "att0_b", "att2_w", "att2_b", "proj_w", "proj_b")]


class FilmDims(C.Structure):
    _fields_ = [("feature_dim", C.c_int32), ("hidden", C.c_int32), ("out_dim", C.c_int32)]


class FilmPtrs(C.Structure):   # mst_film_weights / mst_film_grads
    _fields_ = [(k, C.c_void_p) for k in ("mlp0_w", "mlp0_b", "mlp3_w", "mlp3_b", "head_w", "head_b")]


SYMBOLS = {
    "mst_version": (C.c_int, []),
    "mst_last_error": (C.c_char_p, []),
    "mst_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int]),
    "mst_plan_destroy": (None, [C.c_void_p]),
    "mst_plan_frames": (C.c_int, [C.c_void_p, C.c_int]),
    "mst_plan_feature_dim": (C.c_int, [C.c_void_p]),
    "mst_melfeat_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "mst_melfeat_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    "mst_melfeat_forward_stems": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_longlong, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_melfeat_forward_pcm16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_melfeat_forward_stems_pcm16": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_longlong, C.c_int, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_plan_layout_supported": (C.c_int, [C.c_void_p, C.c_int]),
    "mst_melfeat_forward_io": (C.c_int, [C.c_void_p, C.POINTER(MelfeatIO), C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "mst_encoder_layout_supported": (C.c_int, [C.c_void_p, C.c_int]),
    "mst_encoder_forward_in": (C.c_int, [C.c_void_p, C.POINTER(LogmelIn), C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                         C.POINTER(EncoderTaps), C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_encoder_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(EncoderConfig), C.POINTER(EncoderWeights)]),
    "mst_encoder_destroy": (None, [C.c_void_p]),
    "mst_encoder_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "mst_encoder_set_train_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "mst_encoder_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "mst_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                      C.POINTER(EncoderTaps), C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_encoder_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "mst_encoder_forward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                            C.POINTER(EncoderTrainTaps), C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_dropout_mask": (C.c_int, [C.c_float, C.c_uint64, C.c_longlong, C.c_void_p, C.c_void_p]),
    "mst_head_save_bytes": (C.c_size_t, [C.POINTER(HeadDims), C.c_int, C.c_float]),
    "mst_head_backward_workspace_bytes": (C.c_size_t, [C.POINTER(HeadDims), C.c_int]),
    "mst_head_forward_train": (C.c_int, [C.POINTER(HeadDims), C.POINTER(HeadPtrs), C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_float,
                                         C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_head_backward": (C.c_int, [C.POINTER(HeadDims), C.POINTER(HeadPtrs), C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_float,
                                    C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(HeadPtrs), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_film_save_bytes": (C.c_size_t, [C.POINTER(FilmDims), C.c_int]),
    "mst_film_backward_workspace_bytes": (C.c_size_t, [C.POINTER(FilmDims), C.c_int]),
    "mst_film_forward_train": (C.c_int, [C.POINTER(FilmDims), C.POINTER(FilmPtrs), C.c_void_p, C.c_int, C.c_float, C.c_uint64,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_film_backward": (C.c_int, [C.POINTER(FilmDims), C.POINTER(FilmPtrs), C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                    C.POINTER(FilmPtrs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_encoder_train_update_running_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                         C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_encoder_update_params": (C.c_int, [C.c_void_p, C.POINTER(EncoderWeights), C.c_void_p]),
    "mst_encoder_train_layout_supported": (C.c_int, [C.c_void_p, C.c_int]),
    "mst_encoder_forward_train_in": (C.c_int, [C.c_void_p, C.POINTER(LogmelIn), C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                               C.POINTER(EncoderTrainTaps), C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_encoder_train_conv1_wgrad_in": (C.c_int, [C.c_void_p, C.POINTER(LogmelIn), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                   C.c_size_t, C.c_void_p]),
    "mst_encoder_train_backward_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong,
                                                   C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_encoder_train_backward_apply_phase": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong,
                                                         C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p,
                                                         C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_double]),
    "mst_encoder_train_stats_buffer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t),
                                                 C.POINTER(C.c_size_t)]),
    "mst_encoder_train_scale_buffer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "mst_encoder_update_trunk_params": (C.c_int, [C.c_void_p] * 10),
    "mst_encoder_train_conv1_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_size_t, C.c_void_p]),
    "mst_encoder_train_conv2_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_size_t, C.c_void_p]),
    "mst_encoder_train_conv2_dgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_float, C.c_void_p]),
    "mst_aug_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "mst_aug_apply": (C.c_int, [C.POINTER(AugClip), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                C.c_size_t, C.c_void_p]),
    "mst_aug_apply_strided": (C.c_int, [C.POINTER(AugClip), C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_aug_apply_from": (C.c_int, [C.POINTER(AugClip), C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong,
                                     C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_infonce_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mst_infonce_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mst_infonce_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libmst.so next to this file (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise MstError("building libmst.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


def lib():
    """Load libmst.so (must have been built: `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise MstError(f"{LIB_PATH} not found: the HIP extension is not built. Run __graft_entry__.build() "
                               f"(make -C {CSRC}). There is no CPU fallback.")
            h = C.CDLL(LIB_PATH)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(h, name)
                fn.restype, fn.argtypes = res, args
            _lib = h
    return _lib


def check(rc: int, what: str = ""):
    if rc != MST_OK:
        msg = lib().mst_last_error()
        raise MstError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dptr(t):
    """Device (or host) pointer of a contiguous tensor, or NULL."""
    return C.c_void_p(0 if t is None else t.data_ptr())
