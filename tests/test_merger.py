"""Read-level merger: csrc/merger.cpp behind include/ravvent_merge.h against the CPU
restatement oracle/merger_oracle.py (PARITY UNPINNED: Biopython is absent, the reference stores no expected output).
Host code only -- runs without a GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def mg():
    import ravvent_basecaller_amd as rv
    import ravvent_basecaller_amd.merger as merger
    return merger


@pytest.fixture(scope="module")
def mo():
    from oracle import merger_oracle
    return merger_oracle


def _rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(list(alphabet), n))


def _mutate(rng, s, p):
    out = []
    for c in s:
        r = rng.random()
        if r < p / 3:
            continue                                    # deletion
        if r < 2 * p / 3:
            out.append(rng.choice(list("ACGT")))        # substitution
        elif r < p:
            out.append(c); out.append(rng.choice(list("ACGT")))   # insertion
        else:
            out.append(c)
    return "".join(out)


def test_main_pair_of_the_reference(mg, mo):
    """The only example the reference holds (merger.py:253-255); expected values are this build's restatement."""
    s1, s2 = "AGTTCAGCGATCGGATCCGCGTGC", "GAGATTTTATCCGCGTGCTGTTTACG"
    a = mg.local_align(s1, s2, 0)
    assert a[:2] == ("-AG--TTCAGCGATCGGATCCGCGTGC--------", "GAGATTT------T---ATCCGCGTGCTGTTTACG")
    assert abs(a[2] - 10.4) < 1e-9 and a[3:] == (1, 27)
    assert a == tuple(mo.local_align(s1, s2, 0)[0])
    out = mg.Merger().merge([mg.SeqLogitsPair(s1, [0.5] * len(s1)), mg.SeqLogitsPair(s2, [0.7] * len(s2))])
    assert out.seq == "GAGATTTCAGCGATCGGATCCGCGTGCTGTTTACG"
    assert len(out.logits) == len(out.seq) and abs(out.logits[0] - 0.7) < 1e-6 and abs(out.logits[7] - 0.5) < 1e-6
    # scores_id 2 (localds matrix): the plain overlap, no internal gaps
    assert mg.local_align(s1, s2, 2)[:2] == ("AGTTCAGCGATCGGATCCGCGTGC--------", "------GAGATTTTATCCGCGTGCTGTTTACG")


@pytest.mark.parametrize("scores_id", [0, 1, 2])
def test_first_alignment_matches_oracle_on_random_pairs(mg, mo, scores_id):
    rng = np.random.default_rng(100 + scores_id)
    n_none = 0
    for it in range(1500):
        kind = it % 5
        if kind == 0:        # unrelated
            a, b = _rand_seq(rng, rng.integers(1, 26)), _rand_seq(rng, rng.integers(1, 26))
        elif kind == 1:      # true overlap with errors (the merger's case)
            core = _rand_seq(rng, rng.integers(5, 26))
            a = (_rand_seq(rng, rng.integers(0, 12)) + _mutate(rng, core, 0.15))[-25:]
            b = (_mutate(rng, core, 0.15) + _rand_seq(rng, rng.integers(0, 12)))[:25]
        elif kind == 2:      # low complexity: many co-optimal tracebacks
            a, b = _rand_seq(rng, rng.integers(1, 26), "AC"), _rand_seq(rng, rng.integers(1, 26), "AC")
        elif kind == 3:      # homopolymers / repeats
            a = "A" * int(rng.integers(1, 20)) + _rand_seq(rng, rng.integers(0, 6))
            b = _rand_seq(rng, rng.integers(0, 6)) + "A" * int(rng.integers(1, 20))
        else:                # identical and shifted
            a = _rand_seq(rng, rng.integers(2, 26)); b = a[int(rng.integers(0, len(a))):] + _rand_seq(rng, 3)
        if not a or not b:
            continue
        ref = mo.local_align(a, b, scores_id)
        got = mg.local_align(a, b, scores_id)
        if not ref:
            n_none += 1
            assert got is None, (a, b)
            continue
        ra, rb, rs, rbeg, rend = ref[0]
        assert got is not None, (a, b)
        assert got[0] == ra and got[1] == rb, (a, b, ref[0], got)
        assert abs(got[2] - rs) < 1e-9 and got[3] == rbeg and got[4] == rend
        # structural properties of any pairwise2 alignment: both sequences in full, equal length, no double gap
        assert got[0].replace("-", "") == a and got[1].replace("-", "") == b and len(got[0]) == len(got[1])
        assert all(not (x == "-" and y == "-") for x, y in zip(got[0], got[1]))
    assert n_none > 0 or scores_id == 2


@pytest.mark.parametrize("scores_id", [0, 1, 2])
def test_merge_matches_oracle_on_synthetic_reads(mg, mo, scores_id):
    rng = np.random.default_rng(7 + scores_id)
    for rep in range(12):
        read = _rand_seq(rng, 400 + 50 * rep)
        snippets, pos = [], 0
        while pos < len(read):
            ln = int(rng.integers(8, 60))                  # some chunks shorter than the 25-base overlap
            s = _mutate(rng, read[max(0, pos - 25):pos + ln], 0.08 if rep % 2 else 0.0)
            if rep == 5 and len(snippets) == 3:
                s = ""                                      # an empty call in the middle
            snippets.append((s, rng.random(len(s)).astype(np.float32)))
            pos += ln
        ref_seq, ref_lg = mo.merge([(s, list(l)) for s, l in snippets], scores_id)
        out = mg.Merger(scores_id).merge([mg.SeqLogitsPair(s, list(l)) for s, l in snippets])
        assert out.seq == ref_seq
        assert np.array_equal(np.asarray(out.logits, np.float32), np.asarray(ref_lg, np.float32))
        if rep % 2 == 0 and rep != 5:                       # error-free chunks with exact 25-base overlaps give the read back
            assert out.seq == read


def test_no_alignment_branches(mg, mo):
    """merger.py:181-200: before anything has merged the new snippet replaces the read; afterwards the merge stops."""
    P = mg.SeqLogitsPair
    one = lambda s: P(s, [0.5] * len(s))
    # no common letter -> empty alignment list
    assert mg.local_align("AAAA", "CCCC", 0) is None and mo.local_align("AAAA", "CCCC", 0) == []
    out = mg.Merger().merge([one("AAAA"), one("CCCC"), one("CCGG")])
    ref = mo.merge([("AAAA", [0.5] * 4), ("CCCC", [0.5] * 4), ("CCGG", [0.5] * 4)])
    assert out.seq == ref[0] == "CCCCGG"          # "AAAA" is dropped, then CCCC + CCGG merge
    out = mg.Merger().merge([one("ACGTACGT"), one("CGTACGTT"), one("GGGGGGGG"), one("ACGT")])
    ref = mo.merge([("ACGTACGT", [0.5] * 8), ("CGTACGTT", [0.5] * 8), ("GGGGGGGG", [0.5] * 8), ("ACGT", [0.5] * 4)])
    assert out.seq == ref[0]
    # a single snippet, and empty snippets
    assert mg.Merger().merge([one("ACGT")]).seq == "ACGT"
    assert mg.Merger().merge([one(""), one("ACGT")]).seq == mo.merge([("", []), ("ACGT", [0.5] * 4)])[0] == "ACGT"
    with pytest.raises(ValueError, match="ACGT"):
        mg.Merger(2).merge([one("ACGN"), one("ACGT")])
    with pytest.raises(ValueError):
        mg.Merger(7).merge([one("ACG"), one("ACG")])


def test_reference_helper_classes(mg, mo):
    P = mg.SeqLogitsPair
    assert P.align_logits("A-C", [0.1, 0.2]) == [0.1, -1.0, 0.2] == mo.align_logits("A-C", [0.1, 0.2])
    m = mg.SingleMergerByLogits().merge(P("AC-T", [0.9, 0.1, -1.0, 0.5]), P("-GGT", [-1.0, 0.8, 0.3, 0.5]))
    assert m.seq == "AGGT" and m.logits == [0.9, 0.8, 0.3, 0.5]
    assert (m.seq, m.logits) == mo.single_merge_by_logits("AC-T", "-GGT", [0.9, 0.1, -1.0, 0.5], [-1.0, 0.8, 0.3, 0.5])
    lp = mg.MergerLeftPriority().merge(P("ACG--", [0.5, 0.5, 0.5, -1.0, -1.0]), P("--GTT", [-1.0, -1.0, 0.4, 0.4, 0.4]))
    assert lp.seq == "ACGTT" and lp.logits == [0.5, 0.5, 0.5, 0.4, 0.4]


def test_merge_arrays_takes_the_fused_call_layout(mg, mo):
    """rv_merge_calls reads the [B, L-1] bases / probs / lengths arrays of rv_beam_search_calls directly."""
    rng = np.random.default_rng(3)
    read = _rand_seq(rng, 600)
    B, stride = 40, 47
    bases = np.zeros((B, stride), np.uint8); probs = np.zeros((B, stride), np.float32); lengths = np.zeros(B, np.int32)
    snippets = []
    for i in range(B):
        s = read[max(0, 12 * i - 25):12 * i + 12][:stride]
        lengths[i] = len(s); bases[i, :len(s)] = np.frombuffer(s.encode(), np.uint8)
        probs[i, :len(s)] = rng.random(len(s)).astype(np.float32)
        snippets.append((s, list(probs[i, :len(s)])))
    seq, lg = mg.Merger().merge_arrays(bases, probs, lengths)
    ref_seq, ref_lg = mo.merge(snippets)
    assert seq == ref_seq == read[:12 * B] and np.array_equal(lg, np.asarray(ref_lg, np.float32))


def test_streaming_merger_equals_one_shot(mg, mo):
    """rv_merger_append slab by slab == Merger.merge over all chunks (incl. the early return of merger.py:195-200)."""
    rng = np.random.default_rng(11)
    for case in range(6):
        read = _rand_seq(rng, 700)
        n, stride = 90, 40
        bases = np.zeros((n, stride), np.uint8); probs = rng.random((n, stride)).astype(np.float32); lens = np.zeros(n, np.int32)
        snips = []
        for i in range(n):
            s = _mutate(rng, read[max(0, 7 * i - 25):7 * i + 7], 0.05 * (case % 3))[:stride]
            if case == 4 and i == 50:
                s = "".join("T" if c != "T" else "G" for c in s)[:3] + "NNNNNNNNNN"    # nothing aligns -> merge stops here
            lens[i] = len(s); bases[i, :len(s)] = np.frombuffer(s.encode(), np.uint8)
            snips.append((s, list(probs[i, :len(s)])))
        sm = mg.StreamingMerger()
        for k in range(0, n, 13 + case):
            sm.append(bases[k:k + 13 + case], probs[k:k + 13 + case], lens[k:k + 13 + case])
        seq, lg = sm.result()
        ref_seq, ref_lg = mo.merge(snips)
        assert seq == ref_seq and np.array_equal(lg, np.asarray(ref_lg, np.float32))
        one = mg.Merger().merge_arrays(bases, probs, lens)
        assert one[0] == seq and np.array_equal(one[1], lg)
        sm.close()
