"""Read-level merger with the reference's interface (/root/reference/merger.py) in front of the C++
merger in libravvent_hip.so (`rv_merge_calls`, `rv_local_align`; csrc/merger.cpp, include/ravvent_merge.h).

`Merger(scores_id).merge(list[SeqLogitsPair]) -> SeqLogitsPair` is what the evaluator times as `t_merge`
(/root/reference/ravvent_performance_evaluator.py:73-75).  `Merger.merge_arrays` takes the arrays that
`Basecaller.beam_search_calls` returns (bases / per-base probabilities / lengths per chunk) without building
per-chunk Python objects.  The pairwise alignment the reference gets from Biopython (absent here and un-pinned
there) is restated in C++: SURVEY.md 8f next #1, parity unpinned (oracle/merger_oracle.py)."""
from __future__ import annotations

import ctypes
from typing import List, Sequence

import numpy as np

from . import _capi

_ERR = {-1: "invalid argument", -2: "output buffer too small", -3: "letter outside ACGT with scores_id 2"}


class SeqLogitsPair(object):
    """merger.py:7-37."""

    @classmethod
    def align_logits(cls, seq_gapped: str, logits_non_gapped: List[float]) -> List[float]:
        logits_gapped, index = [], 0
        for c in seq_gapped:
            if c == '-':
                logits_gapped.append(-1.)
            else:
                logits_gapped.append(logits_non_gapped[index])
                index += 1
        return logits_gapped

    @property
    def seq(self) -> str:
        return self._seq

    @property
    def logits(self) -> List[float]:
        return self._logits

    def __init__(self, seq: str, logits) -> None:
        assert len(seq) == len(logits)
        self._seq = seq
        self._logits = logits


class SingleMergerByLogits():
    """merger.py:83-119: per aligned column the base with the higher value; a gap loses."""

    def merge(self, seq_logits_pair1: SeqLogitsPair, seq_logits_pair2: SeqLogitsPair) -> SeqLogitsPair:
        seq1, seq2 = seq_logits_pair1.seq, seq_logits_pair2.seq
        assert len(seq1) == len(seq2)
        seq_merged, logits_merged = [], []
        for n1, n2, l1, l2 in zip(seq1, seq2, seq_logits_pair1.logits, seq_logits_pair2.logits):
            take2 = n1 == '-' or (n2 != '-' and l2 > l1)
            seq_merged.append(n2 if take2 else n1)
            logits_merged.append(l2 if take2 else l1)
        return SeqLogitsPair(seq=''.join(seq_merged), logits=logits_merged)


class MergerLeftPriority():
    """merger.py:39-81 (not used by `Merger`; kept for callers that import it)."""

    def merge(self, seq_logits_pair1: SeqLogitsPair, seq_logits_pair2: SeqLogitsPair) -> SeqLogitsPair:
        seq1, seq2 = seq_logits_pair1.seq, seq_logits_pair2.seq
        assert len(seq1) == len(seq2)
        end = max(i for i, c in enumerate(seq1) if c != '-')          # ValueError when seq1 is all gaps
        seq = seq1[:end + 1] + seq2[end + 1:]
        logits = list(seq_logits_pair1.logits[:end + 1]) + list(seq_logits_pair2.logits[end + 1:])
        return SeqLogitsPair(seq=seq.replace('-', ''), logits=[s for s in logits if s > 0])


def local_align(seq_a: str, seq_b: str, scores_id: int = 0):
    """`pairwise2.align.localms/localds(seq_a, seq_b, <tables of scores_id>)[0]` as (seqA, seqB, score, begin, end),
    or None where the reference's list is empty."""
    lib = _capi.load_library()
    a, b = seq_a.encode(), seq_b.encode()
    cap = len(a) + len(b) + 1
    oa, ob = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(cap)
    n, bg, en, sc = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_double(0)
    rc = lib.rv_local_align(a, len(a), b, len(b), int(scores_id), oa, ob, cap, ctypes.byref(n), ctypes.byref(sc),
                            ctypes.byref(bg), ctypes.byref(en))
    if rc < 0:
        raise ValueError(f"rv_local_align: {_ERR.get(rc, rc)}")
    if rc == 0:
        return None
    return oa.raw[:n.value].decode(), ob.raw[:n.value].decode(), sc.value, bg.value, en.value


class StreamingMerger:
    """`Merger.merge` as a resumable loop (rv_merger_* of include/ravvent_merge.h): append each slab's calls in read
    order -- from a worker thread while the GPU decodes the next slab, ctypes releases the GIL -- then `result()`."""

    def __init__(self, scores_id: int = 0, overlap_seq_len: int = 25):
        self._lib = _capi.load_library()
        self._h = ctypes.c_void_p()
        rc = self._lib.rv_merger_create(int(scores_id), int(overlap_seq_len), ctypes.byref(self._h))
        if rc != 0:
            raise ValueError(f"rv_merger_create: {_ERR.get(rc, rc)}")

    def append(self, bases: np.ndarray, probs: np.ndarray, lengths) -> None:
        bases = np.ascontiguousarray(bases, np.uint8)
        probs = np.ascontiguousarray(probs, np.float32)
        lengths = np.ascontiguousarray(lengths, np.int32)
        n, stride = bases.shape
        assert probs.shape == bases.shape and lengths.shape == (n,)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = self._lib.rv_merger_append(self._h, p(bases), p(probs), p(lengths), stride, n)
        if rc != 0:
            raise ValueError(f"rv_merger_append: {_ERR.get(rc, rc)}")

    def result(self):
        m = ctypes.c_int64(0)
        self._lib.rv_merger_result(self._h, None, None, 0, ctypes.byref(m))
        out_s, out_p = np.empty(max(m.value, 1), np.uint8), np.empty(max(m.value, 1), np.float32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = self._lib.rv_merger_result(self._h, p(out_s), p(out_p), out_s.size, ctypes.byref(m))
        if rc != 0:
            raise ValueError(f"rv_merger_result: {_ERR.get(rc, rc)}")
        return out_s[:m.value].tobytes().decode("ascii"), out_p[:m.value].copy()

    def close(self):
        if self._h:
            self._lib.rv_merger_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Merger():
    """merger.py:121-248."""

    def __init__(self, scores_id=0) -> None:
        self.scores_id = scores_id
        self.overlap_seq_len = 25

    def merge_arrays(self, bases: np.ndarray, probs: np.ndarray, lengths: Sequence[int]):
        """bases u8 [n, stride], probs f32 [n, stride], lengths [n] -> (merged str, f32 array)."""
        lib = _capi.load_library()
        bases = np.ascontiguousarray(bases, np.uint8)
        probs = np.ascontiguousarray(probs, np.float32)
        lengths = np.ascontiguousarray(lengths, np.int32)
        n, stride = bases.shape
        assert probs.shape == bases.shape and lengths.shape == (n,)
        cap = int(lengths.sum()) + 2 * self.overlap_seq_len * n + 1     # every merge adds at most `overlap` gap columns
        out_s, out_p = np.empty(cap, np.uint8), np.empty(cap, np.float32)
        m = ctypes.c_int64(0)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = lib.rv_merge_calls(p(bases), p(probs), p(lengths), stride, n, int(self.scores_id), int(self.overlap_seq_len),
                                p(out_s), p(out_p), cap, ctypes.byref(m))
        if rc != 0:
            raise ValueError(f"rv_merge_calls: {_ERR.get(rc, rc)}")
        return out_s[:m.value].tobytes().decode("ascii"), out_p[:m.value].copy()

    def merge(self, nuc_pred_snippets) -> SeqLogitsPair:
        """merger.py:155-248 through the C++ merge.  The list of pairs becomes the arrays of `merge_arrays` in a handful of vectorised
        numpy calls -- one join of the strings, one concatenation of the per-base values -- not in a Python loop over the chunks
        (round 3: 3.6 us per pair against 1.2 us for the merge itself)."""
        n = len(nuc_pred_snippets)
        seqs = [s.seq for s in nuc_pred_snippets]
        lengths = np.fromiter(map(len, seqs), np.int32, count=n)
        total = int(lengths.sum())
        stride = max(1, int(lengths.max(initial=0)))
        flat_b = np.frombuffer("".join(seqs).encode("ascii"), np.uint8)
        lg = [s.logits for s in nuc_pred_snippets]
        if total and all(isinstance(x, np.ndarray) for x in lg):
            flat_p = np.concatenate(lg).astype(np.float32, copy=False)
        else:                                              # Python lists (the reference's form): one flattening pass
            from itertools import chain
            flat_p = np.fromiter(chain.from_iterable(lg), np.float32, count=total)
        # position of every letter in the padded [n, stride] layout
        starts = np.cumsum(lengths, dtype=np.int64) - lengths
        dst = np.arange(total, dtype=np.int64) + np.repeat(np.arange(n, dtype=np.int64) * stride - starts, lengths)
        bases = np.zeros(n * stride, np.uint8)
        probs = np.zeros(n * stride, np.float32)
        bases[dst] = flat_b
        probs[dst] = flat_p
        seq, out = self.merge_arrays(bases.reshape(n, stride), probs.reshape(n, stride), lengths)
        return SeqLogitsPair(seq=seq, logits=out)
