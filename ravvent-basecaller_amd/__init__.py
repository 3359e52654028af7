"""ravvent-basecaller_amd: MI355X-native (gfx950) inference hot path of the Ravvent basecaller.

One path only: `Basecaller.beam_search_prediction()` /
`greedy_search_prediction()` -- two stacked BiLSTM encoders over raw + event chunks, an LSTM
decoder with Luong / Bahdanau attention and beam search -- as hand-written HIP kernels behind
the C-ABI of include/ravvent_hip.h, with the reference's `Basecaller` class API in front.
There is no CPU fallback: without the HIP library the path raises.
"""
import os as _os

# The asynchronous calls keep several slabs in flight, one HIP stream per slab context.  The ROCm runtime multiplexes a process's
# streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default): with more contexts than queues some share a queue and serialise
# (measured at the C3 shape: depth 4 -> 203 k chunks/s on 4 queues, 262 k on 8; depth 10 -> 278 k on 8 queues, 317 k on 16).  The
# variable is read when the HIP runtime starts, i.e. at the process's first HIP call -- importing this package first is enough;
# an explicit setting wins, RAVVENT_KEEP_ENV=1 leaves the environment alone.  libravvent_hip.so does the same when it is loaded (C callers
# never import this module), and rv_set_option("async_depth", n) returns the warning RV_WQUEUES when n exceeds the setting in force.
if not _os.environ.get("RAVVENT_KEEP_ENV"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .config import RvConfig  # noqa: F401
from . import data_loader, utils, weights, synthetic, dist, evaluator, event_detection, checkpoint  # noqa: F401
from .basecaller import Basecaller  # noqa: F401

__all__ = ["RvConfig", "Basecaller", "data_loader", "utils", "weights", "synthetic", "dist", "evaluator", "event_detection", "checkpoint"]
