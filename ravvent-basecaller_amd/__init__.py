"""ravvent-basecaller_amd: MI355X-native (gfx950) inference hot path of the Ravvent basecaller.

One path only: `Basecaller.beam_search_prediction()` /
`greedy_search_prediction()` -- two stacked BiLSTM encoders over raw + event chunks, an LSTM
decoder with Luong / Bahdanau attention and beam search -- as hand-written HIP kernels behind
the C-ABI of include/ravvent_hip.h, with the reference's `Basecaller` class API in front.
There is no CPU fallback: without the HIP library the path raises.
"""
from .config import RvConfig  # noqa: F401
from . import data_loader, utils, weights, synthetic, dist, evaluator, event_detection, checkpoint  # noqa: F401
from .basecaller import Basecaller  # noqa: F401

__all__ = ["RvConfig", "Basecaller", "data_loader", "utils", "weights", "synthetic", "dist", "evaluator", "event_detection", "checkpoint"]
