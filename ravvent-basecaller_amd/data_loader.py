"""Chunk format + nucleotide tokenizer of the hot path (the `dl.` names the reference's
callers use: /root/reference/ravvent_performance_evaluator.py:95-97).

Format (T2 in SURVEY.md 8a; /root/reference/data_loader.py:12-26,110-126,230-246): a slab is the
3-tuple ``(raw[B,T_r,1] f32, event[B,T_e,5] f32, nuc[B,L] i64)``, post-padded / post-truncated
with ``INPUT_PADDING``; tokens are ``$ ... ^`` with pad id 0.  The chunker that cuts reads into
such slabs is a host-side "next" row and is not part of this module yet.
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
    return [row[row != 0].tobytes().decode("ascii") for row in codes]


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
