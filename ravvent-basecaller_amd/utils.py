"""Hot-path helpers the reference's callers take from its `utils` module
(/root/reference/ravvent_performance_evaluator.py:53,66)."""
from __future__ import annotations

import numpy as np


def input_mask(input_sequence, padding_value):
    """/root/reference/utils.py:26-32 -- True where EVERY feature of a timestep differs from the
    padding value.  (On the GPU path this is folded into the encoder kernels; this host form
    exists for callers and tests.)"""
    return np.all(np.asarray(input_sequence) != padding_value, axis=-1)


def unpack_data_to_input_target(data, input_data_type):
    """/root/reference/utils.py:34-43"""
    raw_sequence, events_sequence, target_sequence = data
    if input_data_type == "raw":
        return raw_sequence, target_sequence
    if input_data_type == "event":
        return events_sequence, target_sequence
    if input_data_type == "joint":
        return (raw_sequence, events_sequence), target_sequence
    raise ValueError(f"input_data_type {input_data_type!r}")


def calc_prob_logits_beam_search_scores(beam_scores):
    """/root/reference/utils.py:123-128 -- exp(score_t - score_{t-1}), score_{-1} = 0, along the
    last axis.  Accepts numpy arrays or torch tensors (host or device); returns the same kind."""
    try:
        import torch
        if isinstance(beam_scores, torch.Tensor):
            prev = torch.zeros_like(beam_scores)
            prev[..., 1:] = beam_scores[..., :-1]
            return torch.exp(beam_scores - prev)
    except ImportError:  # pragma: no cover
        pass
    s = np.asarray(beam_scores)
    prev = np.zeros_like(s)
    prev[..., 1:] = s[..., :-1]
    return np.exp(s - prev)
