"""CPU tests of the host pre-processing row: the C++ event detector against
fixtures produced by THE REFERENCE's own event_detector.py (tests/golden/make_event_golden.py),
and the restated chunker against fixtures produced by THE REFERENCE's own data_loader.prepare_snippets
(tests/golden/make_chunk_golden.py: the module imports with inert stubs for its tensorflow / keras lines)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.mark.parametrize("name", ["events_w6_9", "events_w3_6", "events_w6_6"])
def test_event_detector_matches_reference_output(rv, name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    det = rv.event_detection.EventDetector(window_length1=int(g["w"][0]), window_length2=int(g["w"][1]))
    st, ln, mu, sd = det.run_arrays(g["signal"])
    assert len(st) == len(g["start"])
    assert (st == g["start"]).all() and (ln == g["length"]).all()          # integer work: bit-exact
    assert np.abs(mu - g["mean"]).max() < 1e-9 and np.abs(sd - g["stdv"]).max() < 1e-9
    ev = det.run(g["signal"])
    assert ev[3].end == ev[3].start + ev[3].length == ev[4].start         # events tile the signal


def test_event_detector_edge_cases(rv):
    det = rv.event_detection.EventDetector(6, 9)
    assert det.run(np.zeros(0)) == []
    # flat signal: the reference itself emits one start-up event (0, 2) while its uint32 sample clock
    # is still wrapped (event_detector.py:72-73,281-283) -- reproduced, not "fixed"
    flat = det.run(np.full(500, 7.0))
    assert [(e.start, e.length) for e in flat] == [(0, 2)] and flat[0].mean == 7.0
    steps = det.run(np.r_[np.zeros(100), np.ones(100) * 50, np.zeros(100)] + 0.01 * np.arange(300) % 3)
    assert [(e.start, e.length) for e in steps] == [(0, 4), (4, 95), (99, 91), (190, 9)]   # = reference output


def test_fitting_event_ranges(rv):
    lens = np.full(40, 9)                       # 22 events of 9 samples fit 200
    r = rv.data_loader.compute_fitting_event_ranges(lens, 6, raw_max_len=200)
    assert r[0].tolist() == [0, 22] and r[1].tolist() == [6, 28] and (np.diff(r[:, 0]) == 6).all()
    assert ((r[:, 1] - r[:, 0]) * 9 <= 200).all()
    assert len(rv.data_loader.compute_fitting_event_ranges(np.full(10, 9), 6)) == 0   # never overflows -> no window


def test_chunker_invariants(rv):
    sig, lab = rv.synthetic.make_read(1500, seed=11)
    ranges, syms = lab[:, :2].astype(int), lab[:, 2]
    raw_s, ev_s, nuc_s = rv.data_loader.prepare_snippets(sig, ranges, syms, stride=6)
    assert len(raw_s) == len(ev_s) == len(nuc_s) > 100
    assert max(len(x) for x in raw_s) <= 200 and all(x.shape[1] == 1 for x in raw_s)
    assert all(x.shape[1] == 5 for x in ev_s)
    assert all(s[0] == "$" and s[-1] == "^" and set(s[1:-1]) <= set("ACGT") for s in nuc_s)
    ref = "".join(syms)
    assert all(s[1:-1] in ref for s in nuc_s)                             # each target is a run of the read
    raw, ev, nuc = rv.data_loader.snippets_to_slab(raw_s, ev_s, nuc_s)
    assert raw.shape[1:] == (200, 1) and ev.shape[1:] == (30, 5) and raw.dtype == np.float32 and nuc.dtype == np.int64
    assert (nuc[:, 0] == 2).all() and (nuc.max(axis=1) <= 6).all()
    # padding is the post-padding zero the attention mask keys on (data_loader.py:110-111, utils.py:32)
    k = len(raw_s[0])
    assert (raw[0, k:] == 0).all() and np.allclose(raw[0, :k, 0], raw_s[0][:, 0].astype(np.float32))


@pytest.mark.parametrize("name", ["chunks_a", "chunks_b", "chunks_c"])
def test_chunker_matches_reference_fixtures(rv, name):
    """data_loader.prepare_snippets / compute_fitting_event_ranges against the output of the REFERENCE's own chunker
    (/root/reference/data_loader.py:29-46,70-108; fixtures by tests/golden/make_chunk_golden.py): window ranges and target
    strings exact, standard-scaled raw / event features <= 1e-12.  Pins the chunk format T2 and 8f #2."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    dl = rv.data_loader
    syms = np.array(list(str(g["label_bases"])), dtype=object)
    raw_s, ev_s, tgt = dl.prepare_snippets(g["signal"], g["label_ranges"], syms, int(g["stride"]))
    assert [len(s) for s in raw_s] == g["raw_lens"].tolist()          # the chunk ranges, bit-exact
    assert [len(s) for s in ev_s] == g["ev_lens"].tolist()
    assert tgt == [str(t) for t in g["targets"]]
    assert np.abs(np.concatenate(raw_s).ravel() - g["raw_concat"]).max() < 1e-12
    assert np.abs(np.concatenate(ev_s) - g["ev_concat"]).max() < 1e-12
    fit = dl.compute_fitting_event_ranges(g["fit_lens"].copy(), int(g["stride"]), raw_max_len=dl.MAX_RAW_LEN)
    assert np.array_equal(np.asarray(fit, np.int64), g["fit_ranges"])
    # and the padded slab form of data_loader.py:120-124
    raw, ev, nuc = dl.snippets_to_slab(raw_s, ev_s, tgt)
    assert raw.shape == (len(raw_s), 200, 1) and raw.dtype == np.float32 and ev.shape == (len(raw_s), 30, 5) and nuc.dtype == np.int64
    assert nuc[0, 0] == dl.NUC_TOKEN_START and (nuc[np.arange(len(tgt)), [len(t) - 1 for t in tgt]] == dl.NUC_TOKEN_END).all()
