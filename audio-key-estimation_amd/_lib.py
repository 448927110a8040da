"""ctypes binding of ``libake_hip.so`` (C ABI: ``include/ake_hip.h``).

There is no CPU fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libake_hip.so")
if os.environ.get("AKE_USE_DIAG_LIB") == "1":      # kernel experiments only (tools/, tests/tools/): the -DAKE_DIAG build, `AKE_DIAG=1 csrc/build.sh`
    LIB_PATH = os.path.join(_HERE, "libake_hip_diag.so")

AKE_OK = 0


class AkeError(RuntimeError):
    pass


class CqtConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("hop_length", C.c_int), ("n_bins", C.c_int), ("bins_per_octave", C.c_int),
                ("fmin", C.c_double), ("q_mode", C.c_int), ("decim_half_len", C.c_int), ("decim_beta", C.c_double), ("engine", C.c_int)]


class PcnetConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("pitches", "pitch_classes", "num_layers", "kernel_size", "conv_layers", "n_filters",
                                       "head_layers", "time_pool_size", "genre", "max_pool", "resblock", "denseblock",
                                       "stay_sixth", "only_semitones", "p2pc_conv", "pc2p_mem", "local", "precision")]


# name -> (restype, argtypes); every symbol include/ake_hip.h declares
_P, _I, _I64, _SZ, _F = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.POINTER(C.c_float)
SYMBOLS = {
    "ake_version": (_I, []),
    "ake_last_error": (C.c_char_p, []),
    "ake_build_has_diag": (_I, []),
    "ake_pcnet_precision": (_I, [_P]),
    "ake_cqt_default_config": (_I, [C.POINTER(CqtConfig), _I, _I, _I]),
    "ake_cqt_plan_create": (_I, [C.POINTER(CqtConfig), C.POINTER(_P)]),
    "ake_cqt_plan_destroy": (None, [_P]),
    "ake_cqt_plan_n_bins": (_I, [_P]),
    "ake_cqt_plan_hop": (_I, [_P]),
    "ake_cqt_num_frames": (_I64, [_P, _I64]),
    "ake_cqt_workspace_bytes": (_SZ, [_P, _I, _I64]),
    "ake_cqt_logmag_f32": (_I, [_P, _P, _I, _I64, _I64, _P, _I64, _P, _SZ, _P]),
    "ake_cqt_logmag_ragged_f32": (_I, [_P, _P, _I, _I64, _I64, _P, _P, _I64, _P, _SZ, _P]),
    "ake_cqt_frames_major_supported": (_I, [_P]),
    "ake_cqt_logmag_frames_major_f32": (_I, [_P, _P, _I, _I64, _I64, _P, _P, _SZ, _P]),
    "ake_pcnet_default_config": (_I, [C.POINTER(PcnetConfig), _I, _I]),
    "ake_pcnet_create": (_I, [C.POINTER(PcnetConfig), C.POINTER(_P)]),
    "ake_pcnet_destroy": (None, [_P]),
    "ake_pcnet_pitches": (_I, [_P]),
    "ake_pcnet_num_tensors": (_I, [_P]),
    "ake_pcnet_tensor_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_I64), C.POINTER(_I)]),
    "ake_pcnet_set_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(_I64), _I]),
    "ake_pcnet_finalize": (_I, [_P]),
    "ake_pcnet_workspace_bytes": (_SZ, [_P, _I, _I]),
    "ake_pcnet_forward_f32": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "ake_pcnet_local_frames": (_I, [_P, _I, C.POINTER(_I), C.POINTER(_I)]),
    "ake_pcnet_forward_local_f32": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _SZ, _P]),
    "ake_pcnet_accepts_frames_major": (_I, [_P, _I, _I]),
    "ake_pcnet_forward_frames_major_f32": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "ake_pcnet_num_bn": (_I, [_P]),
    "ake_pcnet_bn_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_I), C.POINTER(_I)]),
    "ake_pcnet_train_workspace_bytes": (_SZ, [_P, _I, _I]),
    "ake_pcnet_forward_train_f32": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "ake_pcnet_grad_floats": (_SZ, [_P]),
    "ake_pcnet_grad_offset": (_I64, [_P, C.c_char_p]),
    "ake_pcnet_backward_f32": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _SZ, _P]),
    "ake_pcnet_load_from_device_f32": (_I, [_P, _P, _P]),
    "ake_pcnet_load_for_training_f32": (_I, [_P, _P, _P]),
    "ake_pcnet_update_running_stats_f32": (_I, [_P, _P, _P, C.c_float, _P]),
    "ake_pcnet_update_recomputed_running_stats_f32": (_I, [_P, _P, _P, C.c_float, _P, _P]),
    "ake_adam_step_f32": (_I, [_P, _P, _P, _P, _P, _SZ] + [C.c_float] * 5 + [_I, C.c_float, _P]),
    "ake_general_step_f32": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _I] + [C.c_float] * 3 + [_I, _P, _P, _P, _P, _P]),
    "ake_pcnet_tap_info": (_I, [_P, C.c_char_p, _I, _I, C.POINTER(_I64)]),
    "ake_debug_keep_taps": (_I, [_I]),
    "ake_pcnet_tap_copy": (_I, [_P, C.c_char_p, _I, _I, _P, _P, _P]),
    "ake_pipeline_workspace_bytes": (_SZ, [_P, _P, _I, _I64]),
    "ake_pipeline_forward_f32": (_I, [_P, _P, _P, _I, _I64, _I64, _P, _P, _P, _P, _SZ, _P]),
    "ake_pipeline_forward_ragged_f32": (_I, [_P, _P, _P, _I, _I64, _I64, _P, _P, _P, _P, _P, _SZ, _P]),
    "ake_resampler_create": (_I, [_I, _I, C.POINTER(_P)]),
    "ake_resampler_destroy": (None, [_P]),
    "ake_resampler_out_len": (_I64, [_P, _I64]),
    "ake_resample_f32": (_I, [_P, _P, _I, _I, _I64, _I64, _I64, _I, _P, _P, _I64, _P, _P]),
    "ake_prof_enable": (_I, [C.c_char_p, _I]),
    "ake_prof_collect": (_I, []),
    "ake_prof_reset": (_I, []),
    "ake_prof_num_entries": (_I, []),
    "ake_prof_entry": (_I, [_I, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_I64)]),
}

_lib = None


def lib():
    """The loaded library; raises if it has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AkeError(f"{LIB_PATH} is missing: build it with audio-key-estimation_amd/csrc/build.sh "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)          # AttributeError if the ABI and this table disagree
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc: int, what: str = ""):
    if rc != AKE_OK:
        msg = lib().ake_last_error()
        raise AkeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def prof_enable(name_filter: str = "", on: bool = True):
    check(lib().ake_prof_enable(name_filter.encode(), 1 if on else 0), "ake_prof_enable")


def prof_results(reset: bool = True):
    """{kernel name: (total_ms, launches)} of everything recorded since the last reset."""
    L = lib()
    check(L.ake_prof_collect(), "ake_prof_collect")
    out = {}
    for i in range(L.ake_prof_num_entries()):
        name, ms, n = C.c_char_p(), C.c_double(), C.c_int64()
        check(L.ake_prof_entry(i, C.byref(name), C.byref(ms), C.byref(n)), "ake_prof_entry")
        out[name.value.decode()] = (ms.value, n.value)
    if reset:
        check(L.ake_prof_reset(), "ake_prof_reset")
    return out
