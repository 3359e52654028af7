"""Regenerates tests/golden/events_*.npz by running THE REFERENCE's own event detector
(/root/reference/event_detection/event_detector.py imports as-is: numpy + math only) on seeded
synthetic reads.  Runs in the build container only; the fixtures (inputs + the reference's
outputs) are what travels.

    python tests/golden/make_event_golden.py
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ravvent_basecaller_amd as rv          # noqa: E402

REF = "/root/reference/event_detection/event_detector.py"
CASES = {"events_w6_9": (6, 9, 400, 1), "events_w3_6": (3, 6, 250, 2), "events_w6_6": (6, 6, 150, 3)}


def main():
    spec = importlib.util.spec_from_file_location("ref_event_detector", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (w1, w2, n_bases, seed) in CASES.items():
        signal, _ = rv.synthetic.make_read(n_bases, seed=seed)
        ev = ref.EventDetector(window_length1=w1, window_length2=w2).run(signal)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), signal=signal.astype(np.int32), w=np.array([w1, w2]),
                            start=np.array([e.start for e in ev], np.int64), length=np.array([e.length for e in ev], np.int64),
                            mean=np.array([e.mean for e in ev]), stdv=np.array([e.stdv for e in ev]))
        print(name, signal.size, "samples ->", len(ev), "events")


if __name__ == "__main__":
    main()
