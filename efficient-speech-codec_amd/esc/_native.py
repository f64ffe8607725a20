"""ctypes binding of libescx.so (the C ABI declared in include/escx.h).

The library is the product: there is no Python/CPU fallback.  If it is missing or no HIP device is visible the
import of this module (or the first call) fails loudly.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_int16, c_int32, c_int64, c_void_p

_LIB_TAG = os.environ.get("ESCX_LIB_TAG", "")          # tuning builds only (build.py ESCX_BUILD_TAG); the product library has no tag
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libescx" + ("_" + _LIB_TAG if _LIB_TAG else "") + ".so")
MAX_SCALES = 8


class EscxConfig(Structure):
    _fields_ = [
        ("in_dim", c_int32), ("in_freq", c_int32), ("n_scales", c_int32), ("h_dims", c_int32 * MAX_SCALES),
        ("max_streams", c_int32), ("win_length", c_int32), ("hop_length", c_int32), ("patch_f", c_int32),
        ("patch_t", c_int32), ("swin_heads", c_int32 * MAX_SCALES), ("swin_depth", c_int32), ("window_size", c_int32),
        ("mlp_ratio", c_float), ("overlap", c_int32), ("group_size", c_int32), ("codebook_size", c_int32),
        ("codebook_dims", c_int32 * MAX_SCALES), ("l2norm", c_int32),
    ]


class EscxDiscConfig(Structure):
    _fields_ = [("sample_rate", c_int32), ("n_rates", c_int32), ("n_periods", c_int32), ("periods", c_int32 * 8), ("n_ffts", c_int32),
                ("fft_sizes", c_int32 * 8), ("n_bands", c_int32), ("bands", (c_float * 2) * 8)]


# name -> (restype, argtypes); every symbol include/escx.h declares
SIGNATURES = {
    "escx_last_error": (c_char_p, []),
    "escx_version": (c_char_p, []),
    "escx_create": (c_int, [POINTER(EscxConfig), c_int, POINTER(c_void_p)]),
    "escx_destroy": (None, [c_void_p]),
    "escx_set_param": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    "escx_finalize_params": (c_int, [c_void_p]),
    "escx_num_required_keys": (c_int, [c_void_p]),
    "escx_required_key": (c_char_p, [c_void_p, c_int]),
    "escx_reserve": (c_int, [c_void_p, c_int, c_int]),
    "escx_workspace_bytes": (c_int64, [c_void_p]),
    "escx_set_precision": (c_int, [c_void_p, c_int]),
    "escx_get_precision": (c_int, [c_void_p]),
    "escx_encode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, POINTER(c_int), POINTER(c_int), c_void_p]),
    "escx_decode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "escx_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "escx_forward_feat": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "escx_num_frames": (c_int, [c_void_p, c_int]),
    "escx_output_samples": (c_int, [c_void_p, c_int]),
    "escx_spec_transform": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "escx_audio_reconstruct": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "escx_patch_embed": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "escx_transformer_layer": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, POINTER(c_int), c_void_p]),
    "escx_pvq_encode": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "escx_pvq_decode": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "escx_patch_deembed": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "escx_profile_enable": (c_int, [c_void_p, c_int]),
    "escx_profile_report": (c_char_p, [c_void_p]),
    "escx_debug_mlp_trace": (c_int, [c_void_p]),
    "escx_test_math": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "escx_test_fastdiv": (c_int, [c_int, c_int]),
    "escx_test_copy_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "escx_codes_pack10": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "escx_codes_unpack10": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "escx_codes_narrow": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "escx_codes_widen": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "escx_flat_param_count": (c_int, [c_void_p]),
    "escx_flat_param_key": (c_char_p, [c_void_p, c_int]),
    "escx_flat_param_offset": (c_int64, [c_void_p, c_int]),
    "escx_flat_param_numel": (c_int64, [c_void_p, c_int]),
    "escx_flat_param_total": (c_int64, [c_void_p]),
    "escx_load_flat_params": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "escx_train_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "escx_train_forward_feat": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "escx_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "escx_train_tape_bytes": (c_int64, [c_void_p]),
    "escx_train_tape_generation": (c_int64, [c_void_p]),
    "escx_stft_loss": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "escx_mel_loss": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "escx_scale_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "escx_grad_norm_clip": (c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    "escx_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_double, c_double, c_double, c_double, c_double, c_void_p, c_void_p]),
    "escx_disc_create": (c_int, [POINTER(EscxDiscConfig), c_int, POINTER(c_void_p)]),
    "escx_disc_destroy": (None, [c_void_p]),
    "escx_disc_param_count": (c_int, [c_void_p]),
    "escx_disc_param_key": (c_char_p, [c_void_p, c_int]),
    "escx_disc_param_offset": (c_int64, [c_void_p, c_int]),
    "escx_disc_param_numel": (c_int64, [c_void_p, c_int]),
    "escx_disc_param_total": (c_int64, [c_void_p]),
    "escx_disc_num_fmaps": (c_int, [c_void_p, c_int]),
    "escx_disc_fmap_shape": (c_int, [c_void_p, c_int, c_int] + [POINTER(c_int)] * 7),
    "escx_disc_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, POINTER(c_void_p), c_void_p]),
    "escx_disc_backward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p, c_void_p]),
    "escx_gan_term": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p]),
    "escx_gan_term_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "escx_disc_set_precision": (c_int, [c_void_p, c_int]),
    "escx_disc_get_precision": (c_int, [c_void_p]),
    "escx_disc_profile_enable": (c_int, [c_int]),
    "escx_disc_profile_report": (c_char_p, []),
    "escx_set_rccl_library": (c_int, [c_char_p]),
    "escx_allgather_codes": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
}

PRECISIONS = {"fp32": 0, "f16x2": 2, "bf16x3": 3}          # include/escx.h ESCX_PRECISION_*
ESCX_ERR_INVALID_ARG, ESCX_ERR_UNSUPPORTED, ESCX_ERR_HIP, ESCX_ERR_STATE, ESCX_ERR_ASSERT = -1, -2, -3, -4, -5

_lib = None
HW_QUEUES_ENV = "GPU_MAX_HW_QUEUES"
HW_QUEUES_WANTED = "8"


def ensure_hw_queues() -> str:
    """The whole-path calls run batch halves on two HIP streams next to torch's and RCCL's.  With ROCm's default of 4 hardware queues
    two of them land on one queue and the overlap turns into serialisation (measured 5420 -> 3900 audio-s/s with an RCCL communicator
    alive).  The variable is only read when HIP initialises, so it is set here - every entry into the library goes
    through load() - unless the user chose a value; if HIP is already up without it, say so instead of silently running slow.
    Returns the value in effect ("default" when HIP initialised before anything set it)."""
    cur = os.environ.get(HW_QUEUES_ENV)
    if cur:
        return cur
    late = False
    try:
        import sys
        tc = sys.modules.get("torch")
        late = bool(tc is not None and tc.cuda.is_initialized())
    except Exception:                                      # pragma: no cover - torch absent or half-imported
        late = False
    if late:
        import warnings
        warnings.warn(f"{HW_QUEUES_ENV} was not set before HIP initialised: the two-stream overlap of esc.ESC.encode/decode may serialise "
                      f"next to RCCL (export {HW_QUEUES_ENV}={HW_QUEUES_WANTED}, or import esc before the first CUDA call)", RuntimeWarning, stacklevel=3)
        return "default"
    os.environ[HW_QUEUES_ENV] = HW_QUEUES_WANTED
    return HW_QUEUES_WANTED


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Loads libescx.so once and types every entry point.  Raises ImportError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    ensure_hw_queues()
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} not found: the HIP library is the only implementation of esc.ESC.encode/decode. "
            "Build it with `python efficient-speech-codec_amd/build.py` (hipcc, gfx950).")
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    """Maps an escx_status to the exception the reference raises in the same situation."""
    if rc == 0:
        return
    msg = (load().escx_last_error() or b"").decode("utf-8", "replace")
    if rc == ESCX_ERR_ASSERT:
        raise AssertionError(msg)
    if rc == ESCX_ERR_INVALID_ARG:
        raise ValueError(msg)
    if rc == ESCX_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(f"escx error {rc}: {msg}")
