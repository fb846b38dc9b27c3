"""ctypes binding of ``libcss_hip.so`` (the C ABI declared in ``include/css_hip.h``).

There is deliberately no CPU fallback here: if the shared library is missing,
or no HIP device is present when a compute entry point is called, the caller
gets a ``RuntimeError`` carrying ``css_last_error()``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p
from pathlib import Path

_PKG = Path(__file__).resolve().parent
# CSS_HIP_LIB: development aid -- another build of the same library (A/B timing of two source states in one
# GPU session, tools/build_variant.sh); the product always loads the in-tree file.
LIB_PATH = Path(os.environ["CSS_HIP_LIB"]) if os.environ.get("CSS_HIP_LIB") else _PKG / "libcss_hip.so"

CSS_OK = 0
CSS_ERR_INVALID = -1
CSS_ERR_NO_DEVICE = -2
CSS_ERR_HIP = -3
CSS_ERR_OOM = -4
CSS_ERR_STATE = -5
METRIC_IP = 0
METRIC_L2 = 1
MAX_K = 2048   # include/css_hip.h CSS_MAX_K


class CssError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libcss_hip error {code}: {message}")
        self.code = code


class DevInfo(ctypes.Structure):
    _fields_ = [
        ("name", ctypes.c_char * 128),
        ("gcn_arch", ctypes.c_char * 64),
        ("compute_units", c_int),
        ("wavefront_size", c_int),
        ("hbm_total_bytes", c_int64),
        ("hbm_free_bytes", c_int64),
        ("lds_bytes_per_cu", c_int),
        ("clock_mhz", c_int),
    ]


class EncoderCfg(ctypes.Structure):
    _fields_ = [
        ("num_layers", c_int),
        ("hidden", c_int),
        ("heads", c_int),
        ("ffn", c_int),
        ("vocab", c_int),
        ("max_pos", c_int),
        ("rel_buckets", c_int),
        ("pad_id", c_int),
        ("max_seq_len", c_int),
        ("ln_eps", c_float),
        ("compute", c_int),
    ]


class Tensor(ctypes.Structure):
    _fields_ = [("name", c_char_p), ("data", POINTER(c_float)), ("numel", c_int64)]


# name -> (restype, argtypes); must list every symbol include/css_hip.h declares
# (tests/test_cabi_symbols.py parses the header and checks this table and the .so).
PROTOTYPES = {
    "css_version": (c_char_p, []),
    "css_last_error": (c_char_p, []),
    "css_device_count": (c_int, [POINTER(c_int)]),
    "css_device_info": (c_int, [c_int, POINTER(DevInfo)]),
    "css_index_create": (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    "css_index_free": (c_int, [c_void_p]),
    "css_index_reset": (c_int, [c_void_p]),
    "css_index_reserve": (c_int, [c_void_p, c_int64]),
    "css_index_ntotal": (c_int, [c_void_p, POINTER(c_int64)]),
    "css_index_dim": (c_int, [c_void_p, POINTER(c_int)]),
    "css_index_metric": (c_int, [c_void_p, POINTER(c_int)]),
    "css_index_device": (c_int, [c_void_p, POINTER(c_int)]),
    "css_index_set_shadow": (c_int, [c_void_p, c_int]),
    "css_index_last_flagged": (c_int, [c_void_p, POINTER(c_int64)]),
    "css_index_last_swept": (c_int, [c_void_p, POINTER(c_int64)]),
    "css_index_shadow_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "css_index_set_range_rows": (c_int, [c_void_p, c_int64]),
    "css_index_set_id_base": (c_int, [c_void_p, c_int64]),
    "css_index_set_search_mode": (c_int, [c_void_p, c_int]),
    "css_index_add": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "css_index_add_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "css_index_add_synthetic": (c_int, [c_void_p, c_int64, c_uint64, c_int64, c_int, c_void_p]),
    "css_index_export": (c_int, [c_void_p, c_int64, c_int64, c_void_p]),
    "css_index_search": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "css_index_search_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "css_index_search_masked": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "css_index_search_masked_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                            c_void_p]),
    "css_merge_topk_dev": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "css_merge_topk_packed_dev": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_int,
                                          c_void_p]),
    "css_encoder_create": (c_int, [POINTER(EncoderCfg), c_int, POINTER(c_void_p)]),
    "css_encoder_free": (c_int, [c_void_p]),
    "css_encoder_load_weights": (c_int, [c_void_p, POINTER(Tensor), c_int]),
    "css_encoder_init_synthetic": (c_int, [c_void_p, c_uint64]),
    "css_encoder_export_weight": (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    "css_encoder_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "css_encoder_forward_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "css_encoder_debug_read": (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    "css_encoder_set_attention_range": (c_int, [c_void_p, c_float]),
    "css_mpnet_rel_bucket": (c_int, [c_int, c_int, c_int]),
    "css_tokenizer_create": (c_int, [c_char_p, c_int, POINTER(c_void_p)]),
    "css_tokenizer_free": (c_int, [c_void_p]),
    "css_tokenizer_vocab_size": (c_int, [c_void_p, POINTER(c_int)]),
    "css_tokenizer_encode_batch": (c_int, [c_void_p, c_char_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int]),
    "css_prof_enable": (c_int, [c_int]),
    "css_prof_reset": (c_int, []),
    "css_prof_read": (c_int, [c_char_p, POINTER(c_double), POINTER(c_int64)]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load ``libcss_hip.so`` (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C claude_semantic_search_amd/csrc`. There is no CPU fallback."
        )
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Load
    # torch first when it is installed so that exactly one HIP runtime is mapped
    # in the process whichever of the two is imported first by the application.
    try:  # pragma: no cover - depends on the environment
        import torch  # noqa: F401
    except Exception:
        pass
    handle = ctypes.CDLL(str(LIB_PATH), mode=os.RTLD_NOW | os.RTLD_LOCAL if hasattr(os, "RTLD_NOW") else 0)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            if os.environ.get("CSS_HIP_LIB"):  # an A/B build of an older tree (tools/build_variant.sh): newer entry points absent
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return handle


def last_error() -> str:
    return (lib().css_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != CSS_OK:
        raise CssError(rc, last_error())


def device_count() -> int:
    n = c_int(0)
    check(lib().css_device_count(ctypes.byref(n)))
    return int(n.value)


def device_info(device: int = 0) -> dict:
    info = DevInfo()
    check(lib().css_device_info(device, ctypes.byref(info)))
    return {
        "name": info.name.decode(),
        "gcn_arch": info.gcn_arch.decode(),
        "compute_units": info.compute_units,
        "wavefront_size": info.wavefront_size,
        "hbm_total_bytes": info.hbm_total_bytes,
        "hbm_free_bytes": info.hbm_free_bytes,
        "lds_bytes_per_cu": info.lds_bytes_per_cu,
        "clock_mhz": info.clock_mhz,
    }


def prof_enable(on: bool) -> None:
    check(lib().css_prof_enable(1 if on else 0))


def prof_reset() -> None:
    check(lib().css_prof_reset())


def prof_read(kernel: str):
    ms = c_double(0.0)
    n = c_int64(0)
    check(lib().css_prof_read(kernel.encode(), ctypes.byref(ms), ctypes.byref(n)))
    return float(ms.value), int(n.value)
