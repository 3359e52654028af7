"""RvConfig: the one configuration record of the hot path, mirrored field-for-field by
``struct RvConfig`` in include/ravvent_hip.h.

The reference has no config system -- hyper-parameters are literals in its scripts
(/root/reference/ravvent.py:14-29, ravvent_performance_evaluator.py:91-103); this record
collects exactly the ones `Basecaller.__init__` (/root/reference/basecaller.py:158-206) takes.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

MODE_RAW, MODE_EVENT, MODE_JOINT = 0, 1, 2
ATT_LUONG, ATT_BAHDANAU = 0, 1
MODES = {"raw": MODE_RAW, "event": MODE_EVENT, "joint": MODE_JOINT}
ATTENTIONS = {"luong": ATT_LUONG, "bahdanau": ATT_BAHDANAU}

RAW_FEATURES = 1     # Encoder(..., inputs_features_num=1)  basecaller.py:175
EVENT_FEATURES = 5   # Encoder(..., inputs_features_num=5)  basecaller.py:176


class CRvConfig(ctypes.Structure):
    """ctypes image of ``struct RvConfig`` (include/ravvent_hip.h)."""
    _fields_ = [
        ("enc_units", ctypes.c_int32),
        ("dec_units", ctypes.c_int32),
        ("enc_depth", ctypes.c_int32),
        ("dec_depth", ctypes.c_int32),
        ("mode", ctypes.c_int32),
        ("attention", ctypes.c_int32),
        ("vocab", ctypes.c_int32),
        ("start_token", ctypes.c_int32),
        ("end_token", ctypes.c_int32),
        ("pad_token", ctypes.c_int32),
        ("padding_value", ctypes.c_float),
        ("max_batch", ctypes.c_int32),
        ("max_raw_len", ctypes.c_int32),
        ("max_event_len", ctypes.c_int32),
        ("max_output_len", ctypes.c_int32),
        ("max_beam", ctypes.c_int32),
        ("device", ctypes.c_int32),
    ]


@dataclass
class RvConfig:
    enc_units: int = 128
    dec_units: int = 128
    enc_depth: int = 2
    dec_depth: int = 1
    mode: str = "joint"            # input_data_type in the reference
    attention: str = "luong"
    vocab: int = 7
    start_token: int = 2           # '$'  data_loader.py:25
    end_token: int = 1             # '^'  data_loader.py:24
    pad_token: int = 0             # ''   data_loader.py:26
    padding_value: float = 0.0     # INPUT_PADDING data_loader.py:14
    max_batch: int = 1024          # evaluator slab size, ravvent_performance_evaluator.py:24
    max_raw_len: int = 300
    max_event_len: int = 45
    max_output_len: int = 64
    max_beam: int = 8
    device: int = 0

    def to_c(self) -> CRvConfig:
        return CRvConfig(
            self.enc_units, self.dec_units, self.enc_depth, self.dec_depth,
            MODES[self.mode], ATTENTIONS[self.attention], self.vocab,
            self.start_token, self.end_token, self.pad_token, float(self.padding_value),
            self.max_batch, self.max_raw_len, self.max_event_len, self.max_output_len,
            self.max_beam, self.device)

    def oracle_cfg(self) -> dict:
        """The dict the CPU oracle takes (tests only hand it over; nothing here imports it)."""
        return dict(mode=self.mode, attention_type=self.attention, start_token=self.start_token,
                    end_token=self.end_token, padding_value=self.padding_value)
