"""ctypes front of oracle/libravvent_oracle.so (C restatement, oracle/ravvent_cpu.c).
TEST INFRASTRUCTURE: checker for sizes the numpy oracle is too slow for, and the
``cpu_baseline`` ("port") engine of bench.py.  Never imported by the product package."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libravvent_oracle.so")


class RvoConfig(ctypes.Structure):
    _fields_ = [("enc_depth", ctypes.c_int), ("mode", ctypes.c_int), ("attention", ctypes.c_int),
                ("vocab", ctypes.c_int), ("start_token", ctypes.c_int), ("end_token", ctypes.c_int),
                ("pad_token", ctypes.c_int), ("padding_value", ctypes.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(LIB)
        _lib.rvo_run.restype = ctypes.c_int
        _lib.rvo_run.argtypes = [ctypes.POINTER(RvoConfig)] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + \
                                [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _lib.rvo_weight_count.restype = ctypes.c_size_t
        _lib.rvo_weight_count.argtypes = [ctypes.POINTER(RvoConfig)]
        _lib.rvo_max_threads.restype = ctypes.c_int
    return _lib


def max_threads() -> int:
    return lib().rvo_max_threads()


def run(cfg: dict, enc_depth: int, vocab: int, blob: np.ndarray, raw, ev, W: int, L: int, greedy=False,
        nthreads: int = 0):
    """cfg: the dict RvConfig.oracle_cfg() makes (+ pad_token optional).  blob: flat fp32 weights.
    Returns (tokens [B,S] i32, scores [B,S] f32 | logits [B,S,V] f32)."""
    mode = {"raw": 0, "event": 1, "joint": 2}[cfg["mode"]]
    c = RvoConfig(enc_depth, mode, {"luong": 0, "bahdanau": 1}[cfg["attention_type"]], vocab,
                  cfg["start_token"], cfg["end_token"], cfg.get("pad_token", 0), cfg.get("padding_value", 0.0))
    blob = np.ascontiguousarray(blob, np.float32)
    assert blob.size == lib().rvo_weight_count(ctypes.byref(c)), "weight blob size mismatch"
    raw = None if raw is None or mode == 1 else np.ascontiguousarray(raw, np.float32)
    ev = None if ev is None or mode == 0 else np.ascontiguousarray(ev, np.float32)
    B = (raw if raw is not None else ev).shape[0]
    T_r = raw.shape[1] if raw is not None else 0
    T_e = ev.shape[1] if ev is not None else 0
    steps = max(L - 1, 0)
    tokens = np.zeros((B, steps), np.int32)
    out2 = np.zeros((B, steps, vocab) if greedy else (B, steps), np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    S = lib().rvo_run(ctypes.byref(c), p(blob), p(raw), p(ev), B, T_r, T_e, W, L, int(greedy), p(tokens), p(out2),
                      nthreads)
    if S < 0:
        raise RuntimeError("rvo_run rejected the configuration")
    return tokens[:, :S], out2[:, :S]
