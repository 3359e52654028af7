"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's read-level merger.

Follows /root/reference/merger.py line by line (SeqLogitsPair :7-37, SingleMergerByLogits :83-119,
Merger.merge :155-248, score tables :124-147, overlap 25 :150).  The alignment itself lives in a
third-party dependency that is ABSENT from /root/reference and from this image: Biopython
``Bio.pairwise2`` (``align.localms`` / ``align.localds``, called at merger.py:168-180; no version is
pinned anywhere in the reference).  It is restated here from the published pure-Python algorithm of
Biopython 1.72-1.81 (``_align``, ``_make_score_matrix_fast``, ``_find_start``, ``_recover_alignments``,
``_find_gap_open``, ``_finish_backtrace``, ``_reverse_matrices``, ``_clean_alignments``; the C
extension ``cpairwise2`` only accelerates the matrix fill with the same arithmetic).

PARITY UNPINNED: the reference holds no test, golden vector or recorded output for the merger (its
``__main__`` pair at merger.py:253-255 prints a result that is not stored), and Biopython cannot be
imported here, so ``algns[0]`` -- in particular its tie order -- is checked against nothing but this
restatement.  Only tests/ may import this module; the product is csrc/merger.cpp.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

OVERLAP_SEQ_LEN = 25                      # merger.py:150
SCORES = {                                # merger.py:124-147
    0: dict(match=1.0, mismatch=-1.0, gap_open=-1.0, gap_extend=-0.2),
    1: dict(match=5.0, mismatch=-4.0, gap_open=-3.0, gap_extend=-0.1),
    2: dict(matrix={("A", "A"): 10.0, ("A", "C"): -3.0, ("A", "G"): -1.0, ("A", "T"): -4.0,
                    ("C", "A"): -3.0, ("C", "C"): 9.0, ("C", "G"): -5.0, ("C", "T"): 0.0,
                    ("G", "A"): -1.0, ("G", "C"): -5.0, ("G", "G"): 7.0, ("G", "T"): -3.0,
                    ("T", "A"): -4.0, ("T", "C"): 0.0, ("T", "G"): -3.0, ("T", "T"): 8.0},
            gap_open=-9.0, gap_extend=-2.0),
}
_PRECISION = 1000
MAX_ALIGNMENTS = 1000


def rint(x: float, precision: int = _PRECISION) -> int:
    return int(x * precision + 0.5)


def calc_affine_penalty(length: int, open_: float, extend: float, penalize_extend_when_opening: bool = False) -> float:
    if length <= 0:
        return 0.0
    penalty = open_ + extend * length
    if not penalize_extend_when_opening:
        penalty -= extend
    return penalty


def _make_score_matrix_fast(A: str, B: str, match_fn, open_A, extend_A, open_B, extend_B):
    """pairwise2._make_score_matrix_fast with align_globally=False, penalize_end_gaps=(False, False),
    penalize_extend_when_opening=False -- the configuration ``align.localms`` / ``localds`` select."""
    first_A_gap = calc_affine_penalty(1, open_A, extend_A)
    first_B_gap = calc_affine_penalty(1, open_B, extend_B)
    local_max_score = 0
    lenA, lenB = len(A), len(B)
    score = [[None] * (lenB + 1) for _ in range(lenA + 1)]
    trace = [[None] * (lenB + 1) for _ in range(lenA + 1)]
    for i in range(lenA + 1):
        score[i][0] = 0
    for i in range(lenB + 1):
        score[0][i] = 0
    col_score = [0]
    for i in range(1, lenB + 1):
        col_score.append(calc_affine_penalty(i, 2 * open_B, extend_B))
    for row in range(1, lenA + 1):
        row_score = calc_affine_penalty(row, 2 * open_A, extend_A)
        for col in range(1, lenB + 1):
            nogap_score = score[row - 1][col - 1] + match_fn(A[row - 1], B[col - 1])
            if row == lenA:                                   # not penalize_end_gaps[0]
                row_open = score[row][col - 1]
                row_extend = row_score
            else:
                row_open = score[row][col - 1] + first_A_gap
                row_extend = row_score + extend_A
            row_score = max(row_open, row_extend)
            if col == lenB:                                   # not penalize_end_gaps[1]
                col_open = score[row - 1][col]
                col_extend = col_score[col]
            else:
                col_open = score[row - 1][col] + first_B_gap
                col_extend = col_score[col] + extend_B
            col_score[col] = max(col_open, col_extend)
            best = max(nogap_score, col_score[col], row_score)
            local_max_score = max(local_max_score, best)
            score[row][col] = 0 if best < 0 else best
            row_score_rint, col_score_rint = rint(row_score), rint(col_score[col])
            row_trace = (1 if rint(row_open) == row_score_rint else 0) + (8 if rint(row_extend) == row_score_rint else 0)
            col_trace = (4 if rint(col_open) == col_score_rint else 0) + (16 if rint(col_extend) == col_score_rint else 0)
            best_rint = rint(best)
            t = 2 if rint(nogap_score) == best_rint else 0
            if row_score_rint == best_rint:
                t += row_trace
            if col_score_rint == best_rint:
                t += col_trace
            trace[row][col] = t
    return score, trace, local_max_score


def _find_start(score, best_score):
    starts = []
    for row in range(len(score)):
        for col in range(len(score[0])):
            s = score[row][col]
            if rint(abs(s - best_score)) <= rint(0):
                starts.append((s, (row, col)))
    return starts


def _finish_backtrace(A, B, aliA, aliB, row, col, gap_char="-"):
    if row:
        aliA += A[row - 1::-1]
    if col:
        aliB += B[col - 1::-1]
    if row > col:
        aliB += gap_char * (len(aliA) - len(aliB))
    elif col > row:
        aliA += gap_char * (len(aliB) - len(aliA))
    return aliA, aliB


def _find_gap_open(A, B, aliA, aliB, end, row, col, col_gap, gap_char, score, trace, in_process, gap_fn, target,
                   index, direction, best_score):
    dead_end = False
    target_score = score[row][col]
    for n in range(target):
        if direction == "col":
            col -= 1
            aliA += gap_char
            aliB += B[col:col + 1]
        else:
            row -= 1
            aliA += A[row:row + 1]
            aliB += gap_char
        actual_score = score[row][col] + gap_fn(index, n + 1)
        if score[row][col] == best_score:
            dead_end = True
            break
        if rint(actual_score) == rint(target_score) and n > 0:
            if not trace[row][col]:
                break
            in_process.append((aliA[:], aliB[:], end, row, col, col_gap, trace[row][col]))
        if not trace[row][col]:
            dead_end = True
    return aliA, aliB, row, col, in_process, dead_end


def _recover_alignments(A, B, starts, best_score, score, trace_m, gap_A_fn, gap_B_fn, reverse=False, gap_char="-"):
    lenA, lenB = len(A), len(B)
    tracebacks = []
    in_process = []
    begin = 0
    for start in starts:
        sc, (row, col) = start
        begin = 0
        if (sc, (row - 1, col - 1)) in starts:
            continue
        if sc <= 0:
            continue
        t = trace_m[row][col]
        if t is None:
            continue
        if (t - t % 2) % 4 == 2:
            trace_m[row][col] = 2
        else:
            continue
        end = -max(lenA - row, lenB - col)
        if not end:
            end = None
        col_distance = lenB - col
        row_distance = lenA - row
        aliA = (col_distance - row_distance) * gap_char + A[lenA - 1:row - 1:-1]
        aliB = (row_distance - col_distance) * gap_char + B[lenB - 1:col - 1:-1]
        in_process += [(aliA, aliB, end, row, col, False, trace_m[row][col])]
    while in_process and len(tracebacks) < MAX_ALIGNMENTS:
        dead_end = False
        aliA, aliB, end, row, col, col_gap, trace = in_process.pop()
        while (row > 0 or col > 0) and not dead_end:
            cache = (aliA[:], aliB[:], end, row, col, col_gap)
            if not trace:
                if col and col_gap:
                    dead_end = True
                else:
                    aliA, aliB = _finish_backtrace(A, B, aliA, aliB, row, col, gap_char)
                break
            elif trace % 2 == 1:
                trace -= 1
                if col_gap:
                    dead_end = True
                else:
                    col -= 1
                    aliA += gap_char
                    aliB += B[col:col + 1]
                    col_gap = False
            elif trace % 4 == 2:
                trace -= 2
                row -= 1
                col -= 1
                aliA += A[row:row + 1]
                aliB += B[col:col + 1]
                col_gap = False
            elif trace % 8 == 4:
                trace -= 4
                row -= 1
                aliA += A[row:row + 1]
                aliB += gap_char
                col_gap = True
            elif trace in (8, 24):
                trace -= 8
                if col_gap:
                    dead_end = True
                else:
                    col_gap = False
                    aliA, aliB, row, col, in_process, dead_end = _find_gap_open(
                        A, B, aliA, aliB, end, row, col, col_gap, gap_char, score, trace_m, in_process, gap_A_fn,
                        col, row, "col", best_score)
            elif trace == 16:
                trace -= 16
                col_gap = True
                aliA, aliB, row, col, in_process, dead_end = _find_gap_open(
                    A, B, aliA, aliB, end, row, col, col_gap, gap_char, score, trace_m, in_process, gap_B_fn,
                    row, col, "row", best_score)
            if trace:
                in_process.append(cache + (trace,))
            trace = trace_m[row][col]
            if score[row][col] == best_score:
                dead_end = True
            elif score[row][col] <= 0:
                begin = max(row, col)
                trace = 0
        if not dead_end:
            if not reverse:
                tracebacks.append((aliA[::-1], aliB[::-1], sc, begin, end))
            else:
                tracebacks.append((aliB[::-1], aliA[::-1], sc, begin, end))
    return _clean_alignments(tracebacks)


def _clean_alignments(alignments):
    unique = []
    for a in alignments:
        if a not in unique:
            unique.append(a)
    i = 0
    while i < len(unique):
        seqA, seqB, sc, begin, end = unique[i]
        if end is None:
            end = len(seqA)
        elif end < 0:
            end = end + len(seqA)
        if begin >= end:
            del unique[i]
            continue
        unique[i] = (seqA, seqB, sc, begin, end)
        i += 1
    return unique


_REVERSE_TRACE = {1: 4, 2: 2, 3: 6, 4: 1, 5: 5, 6: 3, 7: 7, 8: 16, 9: 20, 10: 18, 11: 22, 12: 17, 13: 21, 14: 19,
                  15: 23, 16: 8, 17: 12, 18: 10, 19: 14, 20: 9, 21: 13, 22: 11, 23: 15, 24: 24, 25: 28, 26: 26,
                  27: 30, 28: 25, 29: 29, 30: 27, 31: 31, None: None, 0: 0}


def _reverse_matrices(score, trace):
    rs, rt = [], []
    for col in range(len(score[0])):
        rs.append([score[row][col] for row in range(len(score))])
        rt.append([_REVERSE_TRACE[trace[row][col]] for row in range(len(score))])
    return rs, rt


def local_align(A: str, B: str, scores_id: int = 0):
    """``pairwise2.align.localms(A, B, match, mismatch, open, extend)`` (scores_id 0/1) or
    ``align.localds(A, B, matrix, open, extend)`` (scores_id 2): list of (seqA_gapped, seqB_gapped,
    score, begin, end)."""
    if not A or not B:
        return []
    p = SCORES[scores_id]
    if "matrix" in p:
        m = p["matrix"]

        def match_fn(a, b):
            if (a, b) in m:
                return m[(a, b)]
            return m[(b, a)]          # dictionary_match(symmetric=1); KeyError like the reference otherwise
    else:
        match, mismatch = p["match"], p["mismatch"]

        def match_fn(a, b):
            return match if a == b else mismatch
    op, ex = p["gap_open"], p["gap_extend"]

    def gap_fn(index, length):
        return calc_affine_penalty(length, op, ex)

    score, trace, best = _make_score_matrix_fast(A, B, match_fn, op, ex, op, ex)
    starts = _find_start(score, best)
    algns = _recover_alignments(A, B, starts, best, score, trace, gap_fn, gap_fn)
    if not algns:
        score, trace = _reverse_matrices(score, trace)
        starts = [(z, (y, x)) for z, (x, y) in starts]
        algns = _recover_alignments(B, A, starts, best, score, trace, gap_fn, gap_fn, reverse=True)
    return algns


# ------------------------------------------------------------------------------------------ merger.py
def align_logits(seq_gapped: str, logits_non_gapped: Sequence[float]) -> List[float]:
    """SeqLogitsPair.align_logits (merger.py:9-23)."""
    out, index = [], 0
    for c in seq_gapped:
        if c == "-":
            out.append(-1.0)
        else:
            out.append(logits_non_gapped[index])
            index += 1
    return out


def single_merge_by_logits(seq1, seq2, logits1, logits2) -> Tuple[str, List[float]]:
    """SingleMergerByLogits.merge (merger.py:88-119)."""
    assert len(seq1) == len(seq2)
    seq, lg = "", []
    for n1, n2, l1, l2 in zip(seq1, seq2, logits1, logits2):
        if n1 == "-":
            seq += n2
            lg.append(l2)
        elif n2 == "-":
            seq += n1
            lg.append(l1)
        elif l2 > l1:
            seq += n2
            lg.append(l2)
        else:
            seq += n1
            lg.append(l1)
    return seq, lg


def merge(snippets: Sequence[Tuple[str, Sequence[float]]], scores_id: int = 0,
          overlap: int = OVERLAP_SEQ_LEN) -> Tuple[str, List[float]]:
    """Merger.merge (merger.py:155-248): snippets = [(seq, logits)], returns (seq_merged, logits_merged)."""
    seq_merged = snippets[0][0]
    logits_merged = list(snippets[0][1])
    merge_flag = False
    for i in range(1, len(snippets)):
        seq_app, logits_app = snippets[i][0], list(snippets[i][1])
        seq1_ov, seq2_ov = seq_merged[-overlap:], seq_app[:overlap]
        lg1_ov, lg2_ov = logits_merged[-overlap:], logits_app[:overlap]
        algns = local_align(seq1_ov, seq2_ov, scores_id)
        if len(algns) == 0:
            if not merge_flag:
                seq_merged, logits_merged = seq_app, logits_app
                continue
            return seq_merged, logits_merged
        merge_flag = True
        a = algns[0]
        s1g, s2g = a[0], a[1]
        m_seq, m_lg = single_merge_by_logits(s1g, s2g, align_logits(s1g, lg1_ov), align_logits(s2g, lg2_ov))
        seq_merged = seq_merged[:-overlap] + m_seq + seq_app[overlap:]
        logits_merged = logits_merged[:-overlap] + m_lg + logits_app[overlap:]
    return seq_merged, logits_merged
