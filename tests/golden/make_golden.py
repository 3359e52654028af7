"""Regenerates tests/golden/*.npz from the numpy fp64 oracle (oracle/ravvent_oracle.py).

PARITY UNPINNED: the reference holds no vectors for this path and TensorFlow/TFA are not
installed, so these fixtures pin the build's own oracle (and, through it, every backend) rather
than the reference's arithmetic.  Weights and inputs come from the splitmix64 generators
(ravvent-basecaller_amd/weights.py `scheme='hash'`, synthetic.hash_slab), which are exact integer
arithmetic and regenerate bit-identically anywhere; only the generator seeds and the expected
outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ravvent_basecaller_amd as rv          # noqa: E402
from oracle import ravvent_oracle as O       # noqa: E402

CASES = {
    # name: (mode, attention, enc_depth, B, T_r, T_e, W, L, weight_seed, input_seed)
    "joint_luong_w5": ("joint", "luong", 2, 4, 40, 8, 5, 12, 7, 1),
    "joint_bahdanau_w3": ("joint", "bahdanau", 2, 3, 32, 6, 3, 10, 8, 2),
    "raw_luong_greedy": ("raw", "luong", 2, 3, 48, 0, 1, 12, 9, 3),
    "event_luong_w2_d1": ("event", "luong", 1, 5, 0, 12, 2, 9, 10, 4),
}


def build(name):
    mode, att, depth, B, T_r, T_e, W, L, wseed, iseed = CASES[name]
    cfg = rv.RvConfig(mode=mode, attention=att, enc_depth=depth)
    flat = rv.weights.init_weights(cfg, seed=wseed, scheme="hash")
    w = rv.weights.flat_to_nested(cfg, flat)
    raw, ev = rv.synthetic.hash_slab(B, max(T_r, 1), max(T_e, 1), seed=iseed)
    return cfg, flat, w, raw, ev, W, L


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        cfg, flat, w, raw, ev, W, L = build(name)
        taps = {}
        if "greedy" in name:
            tok, logits = O.greedy_search(w, cfg.oracle_cfg(), raw, ev, L, dtype=np.float64, taps=taps)
            extra = dict(logits=logits.astype(np.float32))
        else:
            tok, sc = O.beam_search(w, cfg.oracle_cfg(), raw, ev, W, L, dtype=np.float64, taps=taps)
            extra = dict(scores=sc.astype(np.float32), step_ids=taps["step_ids"], parent_ids=taps["parent_ids"],
                         step_logits=taps["step_logits"].astype(np.float32))
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"), tokens=tok,
            enc_output_sample=taps["enc_output"][:, ::5, ::16].astype(np.float32),
            enc_output_sum=np.array(taps["enc_output"].sum()), mask=taps["mask"],
            weight_checksum=np.array(float(np.sum(rv.weights.pack(cfg, flat).astype(np.float64)))),
            strings=np.array(O.tokens_to_nuc_sequences(tok)), **extra)
        print(name, tok.shape, O.tokens_to_nuc_sequences(tok))


if __name__ == "__main__":
    main()
