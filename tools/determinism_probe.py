"""Diagnostic: are two identical calls bit-identical?  (encoder output and scores, every projection form, a few shapes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ravvent_basecaller_amd as rv
for B, Tr, Te in ((12, 50, 10), (5, 37, 9), (256, 300, 30), (300, 100, 30), (64, 300, 30)):
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B)
    flat = rv.weights.init_weights(bc.cfg, seed=3); flat["b_fc"][1] = 2.5
    bc.set_weights_flat(flat)
    raw, ev, _ = rv.synthetic.make_slab(B, Tr, Te, seed=4)
    for split in (0, 1, 2):
        bc.set_option("split_projection", split)
        ref = None; diff = 0; tapdiff = None
        for rep in range(6):
            tok, sc = bc.beam_search_prediction((raw, ev), 5, 40)
            e = bc.get_tensor("enc_output").copy(); s = sc.numpy().copy()
            if ref is None: ref = (e, s)
            else: diff += int((e != ref[0]).sum()) + int((s != ref[1]).sum())
        bc.set_option("debug_taps", 1)
        tok2, sc2 = bc.beam_search_prediction((raw, ev), 5, 40)
        bc.set_option("debug_taps", 0)
        e2 = bc.get_tensor("enc_output").copy()
        print(f"B={B} T=({Tr},{Te}) split={split}: differing words over 5 repeats {diff}; taps run: enc differs {int((e2 != ref[0]).sum())}, "
              f"scores differ {int((sc2.numpy() != ref[1]).sum())} (max {np.abs(sc2.numpy() - ref[1]).max():.2e})")
    bc.close()
