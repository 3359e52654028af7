"""ctypes binding of libravvent_hip.so (include/ravvent_hip.h).  Fails loudly: there is no
CPU fallback anywhere behind this module."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_void_p

from .config import CRvConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libravvent_hip.so")

RV_OK = 0
ERROR_NAMES = {-1: "RV_EINVAL", -2: "RV_ENOMEM", -3: "RV_EHIP", -4: "RV_ESTATE", -5: "RV_EUNSUPPORTED"}

# every symbol include/ravvent_hip.h declares: (name, restype, argtypes)
_F, _I = POINTER(c_float), POINTER(c_int32)
SYMBOLS = [
    ("rv_abi_version", c_int32, []),
    ("rv_create", c_int32, [POINTER(CRvConfig), POINTER(c_void_p)]),
    ("rv_destroy", None, [c_void_p]),
    ("rv_last_error", c_char_p, [c_void_p]),
    ("rv_weight_count", c_size_t, [c_void_p]),
    ("rv_load_weights", c_int32, [c_void_p, c_void_p, c_size_t]),
    ("rv_beam_search", c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, _I]),
    ("rv_beam_search_dev", c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, _I]),
    ("rv_beam_search_calls", c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, _I]),
    ("rv_beam_search_submit", c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, _I]),
    ("rv_beam_search_collect", c_int32, [c_void_p, c_int32, c_void_p, c_void_p, _I]),
    ("rv_beam_search_submit_dev", c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, _I]),
    ("rv_beam_search_collect_dev", c_int32, [c_void_p, c_int32, _I]),
    ("rv_beam_search_submit_calls", c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, _I]),
    ("rv_beam_search_collect_calls", c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, _I]),
    ("rv_greedy_search", c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, _I]),
    ("rv_greedy_search_dev", c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, _I]),
    ("rv_set_option", c_int32, [c_void_p, c_char_p, c_int32]),
    ("rv_get_tensor", c_int32, [c_void_p, c_char_p, c_void_p, c_size_t, POINTER(c_size_t)]),
    ("rv_get_profile", c_int32, [c_void_p, c_char_p, POINTER(c_double), POINTER(c_int64)]),
    ("rv_profile_names", c_int32, [c_void_p, ctypes.c_char_p, c_size_t]),
    ("rv_reset_profile", c_int32, [c_void_p]),
    ("rv_detect_events", c_int32, [c_void_p, c_size_t, c_int32, c_int32, c_double, c_double, c_double,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, POINTER(c_size_t)]),
    # include/ravvent_merge.h
    ("rv_merge_calls", c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                 c_int64, POINTER(c_int64)]),
    ("rv_merger_create", c_int32, [c_int32, c_int32, POINTER(c_void_p)]),
    ("rv_merger_append", c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32]),
    ("rv_merger_result", c_int32, [c_void_p, c_void_p, c_void_p, c_int64, POINTER(c_int64)]),
    ("rv_merger_destroy", None, [c_void_p]),
    ("rv_local_align", c_int32, [c_char_p, c_int32, c_char_p, c_int32, c_int32, c_char_p, c_char_p, c_int32, _I,
                                 POINTER(c_double), _I, _I]),
]

_lib = None


class RavventHipError(RuntimeError):
    pass


def load_library(path: str | None = None):
    """Load libravvent_hip.so and bind every declared symbol.  Raises if the library has not
    been built (run ``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C
    ravvent-basecaller_amd/csrc``)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RAVVENT_HIP_LIB") or LIB_PATH     # RAVVENT_HIP_LIB: an alternative build of the same ABI (A/B timing)
    if not os.path.exists(p):
        raise RavventHipError(f"{p} not found: build the HIP library first (there is no CPU fallback)")
    try:   # torch bundles its own libamdhip64.so.7; load it first so both share one HIP runtime
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass
    lib = ctypes.CDLL(p)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)       # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(lib, handle, rc: int, what: str):
    if rc != RV_OK:
        msg = lib.rv_last_error(handle)
        raise RavventHipError(f"{what}: {ERROR_NAMES.get(rc, rc)}: {msg.decode() if msg else ''}")
