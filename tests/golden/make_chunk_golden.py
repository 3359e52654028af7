"""Regenerates tests/golden/chunks_*.npz by running THE REFERENCE's own chunker -- `prepare_snippets` and
`compute_fitting_event_ranges` of /root/reference/data_loader.py (numpy + sklearn + the reference's event detector) -- on
seeded synthetic reads.  The reference module's first lines import tensorflow / keras, which are not installed here and
which the chunker itself never calls: those three import lines are satisfied with INERT stub modules
(empty namespaces; `tf.keras.utils.Sequence` is an empty base class for the training generator the module also defines; `Tokenizer` is an attribute bag so that the module-level `nuc_tk = Tokenizer(...)` assignments run;
`pad_sequences` raises if anyone calls it).  Nothing of the reference is copied: the fixtures hold inputs (signal, labels)
and the reference's numeric outputs.  Runs in the build container only; the fixtures are what travels.

    python tests/golden/make_chunk_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ravvent_basecaller_amd as rv          # noqa: E402

REF_DIR = "/root/reference"
CASES = {"chunks_a": dict(n_bases=300, seed=21, mean_dwell=9.0, stride=6),
         "chunks_b": dict(n_bases=420, seed=22, mean_dwell=7.5, stride=6),
         "chunks_c": dict(n_bases=260, seed=23, mean_dwell=11.0, stride=4)}


def _inert(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference_data_loader():
    class _Bag:                                   # stands in for keras' Tokenizer: holds attributes, does nothing
        def __init__(self, *a, **k):
            pass

    def _never(*a, **k):
        raise RuntimeError("inert stub: the chunker under test must not reach tensorflow / keras")
    for name in ("tensorflow.keras.preprocessing", "keras", "keras.preprocessing"):
        _inert(name)
    # `class RawEventNucDataGenerator(tf.keras.utils.Sequence)` (a training feed, out of scope) needs a base class to exist
    utils = _inert("tensorflow.keras.utils", Sequence=type("Sequence", (), {}))
    keras = _inert("tensorflow.keras", utils=utils)
    _inert("tensorflow", keras=keras)
    _inert("tensorflow.keras.preprocessing.sequence", pad_sequences=_never)
    _inert("keras.preprocessing.text", Tokenizer=_Bag)
    sys.path.insert(0, REF_DIR)                   # for `from event_detection.event_detector import EventDetector`
    spec = importlib.util.spec_from_file_location("ref_data_loader", os.path.join(REF_DIR, "data_loader.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def main():
    ref = load_reference_data_loader()
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, c in CASES.items():
        signal, labels = rv.synthetic.make_read(c["n_bases"], seed=c["seed"], mean_dwell=c["mean_dwell"])
        ranges = labels[:, :2].astype(int)
        syms = labels[:, 2]
        raw_s, ev_s, tgt = ref.prepare_snippets(signal, ranges, syms, c["stride"])
        # the range function alone, on lengths that also hit its early-exit branches
        rng = np.random.default_rng(c["seed"])
        lens = rng.integers(3, 40, 180)
        fit = ref.compute_fitting_event_ranges(lens.copy(), c["stride"], raw_max_len=ref.MAX_RAW_LEN)
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            signal=signal.astype(np.int32), label_ranges=ranges.astype(np.int64), label_bases=np.array("".join(syms)),
            stride=np.int64(c["stride"]),
            raw_lens=np.array([len(s) for s in raw_s], np.int64), raw_concat=np.concatenate(raw_s).astype(np.float64).ravel(),
            ev_lens=np.array([len(s) for s in ev_s], np.int64), ev_concat=np.concatenate(ev_s).astype(np.float64),
            targets=np.array(tgt), fit_lens=lens.astype(np.int64), fit_ranges=np.asarray(fit, np.int64))
        print(name, signal.size, "samples ->", len(raw_s), "chunks; raw", min(map(len, raw_s)), "-", max(map(len, raw_s)),
              "events", min(map(len, ev_s)), "-", max(map(len, ev_s)), "target", tgt[0])


if __name__ == "__main__":
    main()
