"""`EventDetector` with the reference's interface
(/root/reference/event_detection/event_detector.py:26-83) in front of the C++ streaming detector
in libravvent_hip.so (`rv_detect_events`, csrc/event_detect.cpp).  Host pre-processing: it sits in
the reference's `t_data_loading`, outside the timed hot path, where the
reference spends seconds per read in a per-sample Python loop."""
from __future__ import annotations

import ctypes
from typing import NamedTuple

import numpy as np

from . import _capi


class Event(NamedTuple):
    start: int
    length: int
    mean: float
    stdv: float

    @property
    def end(self) -> int:
        return self.start + self.length


class EventDetector:
    def __init__(self, window_length1=3, window_length2=6, threshold1=1.4, threshold2=9., peak_height=0.2):
        self.params = {"window_length1": window_length1, "window_length2": window_length2,
                       "threshold1": threshold1, "threshold2": threshold2, "peak_height": peak_height}

    def run_arrays(self, raw):
        """-> (start i64[n], length i64[n], mean f64[n], stdv f64[n])"""
        lib = _capi.load_library()
        x = np.ascontiguousarray(np.asarray(raw), np.float64).ravel()
        cap = x.size + 8    # an event is at least one sample long: this capacity can never be exceeded
        st = np.empty(cap, np.int64); ln = np.empty(cap, np.int64)
        mu = np.empty(cap, np.float64); sd = np.empty(cap, np.float64)
        n = ctypes.c_size_t(0)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = lib.rv_detect_events(p(x), x.size, self.params["window_length1"], self.params["window_length2"],
                                  self.params["threshold1"], self.params["threshold2"], self.params["peak_height"],
                                  p(st), p(ln), p(mu), p(sd), cap, ctypes.byref(n))
        if rc != 0:
            raise _capi.RavventHipError(f"rv_detect_events failed ({rc}); {n.value} events for capacity {cap}")
        k = n.value
        return st[:k], ln[:k], mu[:k], sd[:k]

    def run(self, raw):
        """list of Event(start, length, mean, stdv), as the reference's run() (:75-83)."""
        st, ln, mu, sd = self.run_arrays(raw)
        return [Event(int(a), int(b), float(c), float(d)) for a, b, c, d in zip(st, ln, mu, sd)]
