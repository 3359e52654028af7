"""Chunk format + nucleotide tokenizer of the hot path (the `dl.` names the reference's
callers use: /root/reference/ravvent_performance_evaluator.py:95-97).

Format (T2 in SURVEY.md 8a; /root/reference/data_loader.py:12-26,110-126,230-246): a slab is the
3-tuple ``(raw[B,T_r,1] f32, event[B,T_e,5] f32, nuc[B,L] i64)``, post-padded / post-truncated
with ``INPUT_PADDING``; tokens are ``$ ... ^`` with pad id 0.  The chunker that cuts reads into
such slabs (`prepare_snippets` / `load_data_from_single_signal_label`) is restated at the bottom:
host-side pre-processing, a "next" row of the scope table.  Both its event detector
and the snippet cutter are pinned against the reference's own output (tests/golden/events_*.npz,
tests/golden/chunks_*.npz: `prepare_snippets` / `compute_fitting_event_ranges` of the reference run in
the build container with inert stubs for the module's tensorflow / keras import lines).
"""
from __future__ import annotations

import numpy as np

INPUT_PADDING = 0.0
MAX_RAW_LEN = 200
MAX_EVENT_LEN = 30


class NucTokenizer:
    """Char-level tokenizer with the reference's hand-set vocabulary
    (/root/reference/data_loader.py:20-22; Keras semantics restated in SURVEY.md A.8)."""

    def __init__(self):
        self.word_index = {"": 0, "^": 1, "$": 2, "a": 3, "c": 4, "g": 5, "t": 6}
        self.index_word = {i: w for w, i in self.word_index.items()}

    def texts_to_sequences(self, texts):
        return [[self.word_index[ch] for ch in t.lower() if ch in self.word_index] for t in texts]

    def sequences_to_texts(self, sequences):
        return [" ".join(self.index_word[int(i)] for i in seq if int(i) in self.index_word)
                for seq in np.asarray(sequences)]


nuc_tk = NucTokenizer()
NUC_TOKEN_END = nuc_tk.word_index["^"]
NUC_TOKEN_START = nuc_tk.word_index["$"]
NUC_TOKEN_PAD = nuc_tk.word_index[""]

# ids -> ASCII for the vectorised string path: 0/1/2 ('' ^ $) vanish, 3..6 -> ACGT
_ASCII = np.zeros(256, np.uint8)
_ASCII[3:7] = np.frombuffer(b"ACGT", np.uint8)


def tokens_to_strings(tokens) -> list:
    """Vectorised equivalent of Basecaller.tokens_to_nuc_sequences
    (/root/reference/basecaller.py:289-294): map ids to chars, drop '' ^ $, upper-case."""
    t = np.asarray(tokens)
    if t.ndim == 1:
        t = t[None]
    codes = _ASCII[np.clip(t, 0, 255).astype(np.uint8)]
    keep = codes != 0
    flat = codes[keep].tobytes().decode("ascii")          # one decode for the whole slab
    ends = np.cumsum(keep.sum(axis=1)).tolist()
    return [flat[a:b] for a, b in zip([0] + ends[:-1], ends)]


def pad_sequences(seqs, maxlen=None, dtype="float32", value=0.0):
    """Keras pad_sequences(padding='post', truncating='post'), the only mode
    the reference uses (/root/reference/data_loader.py:110-111,124)."""
    seqs = [np.asarray(s) for s in seqs]
    if maxlen is None:
        maxlen = max((len(s) for s in seqs), default=0)
    tail = seqs[0].shape[1:] if seqs and seqs[0].ndim > 1 else ()
    out = np.full((len(seqs), maxlen) + tuple(tail), value, dtype=dtype)
    for i, s in enumerate(seqs):
        n = min(len(s), maxlen)
        out[i, :n] = s[:n]
    return out


def pad_input_snippets(snippets, maxlen):
    """/root/reference/data_loader.py:110-111"""
    return pad_sequences(snippets, maxlen=maxlen, dtype="float32", value=INPUT_PADDING)


# ----------------------------------------------------------------------------------------------
# Chunker (host pre-processing; /root/reference/data_loader.py:29-126)
ED_WINDOW_LENGTH_1 = 6      # data_loader.py:12
ED_WINDOW_LENGTH_2 = 9      # data_loader.py:13


def _standard_scale_fit(x):
    """sklearn StandardScaler().fit: per-column mean and population std (ddof 0); zero-variance
    columns scale by 1."""
    mean = x.mean(axis=0)
    scale = x.std(axis=0)
    scale = np.where(scale == 0.0, 1.0, scale)
    return mean, scale


def compute_fitting_event_ranges(events_lens, stride, raw_max_len=200):
    """data_loader.py:29-46 -- [start, end) event ranges every `stride` events whose summed lengths
    stay <= raw_max_len; stops when the tail no longer overflows the window."""
    cum = np.cumsum(events_lens, axis=0, dtype=np.int32)
    out = []
    for i in range(0, len(events_lens), stride):
        end_id = int(np.argmax(cum > raw_max_len))
        if end_id == 0:
            break
        out.append((i, end_id))
        if (i + stride - 1) >= len(cum):
            break
        cum = cum - cum[i + stride - 1]
    return np.array(out)


def prepare_snippets(raw, nuc_raw_ranges, nuc_reference_symbols, stride, max_raw_len=MAX_RAW_LEN):
    """data_loader.py:70-108 -- events -> 5 scaled features, raw standard-scaled per read, windows of
    <= max_raw_len samples every `stride` events, target strings '$...^' per window."""
    from .event_detection import EventDetector
    raw = np.asarray(raw)
    nuc_raw_ranges = np.asarray(nuc_raw_ranges)
    st, ln, mu, sd = EventDetector(window_length1=ED_WINDOW_LENGTH_1, window_length2=ED_WINDOW_LENGTH_2).run_arrays(raw)
    dmean = np.concatenate([[0.0], np.diff(mu)]) if len(mu) else np.zeros(0)
    events = np.column_stack([st, st + ln, ln, mu, sd, mu ** 2, dmean]).astype(np.float64)
    ev_mean, ev_scale = _standard_scale_fit(events[:, 2:])
    keep = np.logical_and(events[:, 0] >= nuc_raw_ranges[0, 0], events[:, 1] <= nuc_raw_ranges[-1, 1])
    events = events[keep]
    events[0, 2] += events[0, 0] - nuc_raw_ranges[0, 0]
    events[0, 0] = nuc_raw_ranges[0, 0]
    events[-1, 2] = nuc_raw_ranges[-1, 1] - events[-1, 0]
    r = raw.reshape(-1, 1).astype(np.float64)
    r_mean, r_scale = _standard_scale_fit(r)
    raw_sc = (r - r_mean) / r_scale
    events_ranges = compute_fitting_event_ranges(events[:, 2], stride, raw_max_len=max_raw_len)
    raw_ranges = np.column_stack((events[:, 0][events_ranges[:, 0]].astype(np.int32),
                                  events[:, 0][events_ranges[:, 1] - 1].astype(np.int32)))
    events_sc = (events[:, 2:] - ev_mean) / ev_scale
    raw_snippets = [raw_sc[a:b] for a, b in raw_ranges]
    event_snippets = [events_sc[a:b] for a, b in events_ranges]
    ids_lens = nuc_raw_ranges[:, 1] - nuc_raw_ranges[:, 0]
    id_seq = np.repeat(np.arange(nuc_raw_ranges.shape[0]), ids_lens)
    if nuc_raw_ranges[0, 0] != 0:
        id_seq = np.concatenate((np.full(nuc_raw_ranges[0, 0], -1), id_seq))
    syms = np.asarray(nuc_reference_symbols)
    nuc_sym_snippets = ["$" + "".join(syms[np.unique(id_seq[a:b])]) + "^" for a, b in raw_ranges]
    return raw_snippets, event_snippets, nuc_sym_snippets


def snippets_to_slab(raw_snippets, event_snippets, nuc_sym_snippets, max_raw_len=MAX_RAW_LEN, max_event_len=MAX_EVENT_LEN):
    """data_loader.py:120-124: pad to (raw[n,T_r,1] f32, event[n,T_e,5] f32, nuc[n,L] i64)."""
    raw = pad_input_snippets(raw_snippets, max_raw_len)
    ev = pad_input_snippets(event_snippets, max_event_len)
    nuc = pad_sequences(nuc_tk.texts_to_sequences(nuc_sym_snippets), maxlen=None, value=NUC_TOKEN_PAD, dtype="int64")
    return raw, ev, nuc


def load_data_from_single_signal_label(signal_path, label_path, stride):
    """data_loader.py:113-126"""
    raw = np.loadtxt(signal_path, dtype=int)
    label = np.loadtxt(label_path, dtype=object)
    return snippets_to_slab(*prepare_snippets(raw, label[:, :2].astype(int), label[:, 2], stride))
