"""Diagnostic: per-slab time of the reference's (enc_depth, dec_depth) model families at the C3 shape, persistent vs per-step decode."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = 256, 300, 30, 5, 48
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
gc.disable()
for enc_d, dec_d in ((1, 1), (2, 1), (2, 2), (3, 1), (3, 2)):      # accuracy_results_all.*.json of the reference
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, encoder_depth=enc_d, decoder_depth=dec_d,
                       max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
    bc.init_random_weights(seed=22)
    out = []
    for persist in (1, 0):
        bc.set_option("persistent_decode", persist)
        for _ in range(8):
            bc.beam_search_prediction(x, W, L)
        ts = []
        for _ in range(30):
            t = time.perf_counter(); bc.beam_search_prediction(x, W, L); ts.append(time.perf_counter() - t)
        out.append(np.median(ts) * 1e3)
    S = bc.beam_search_prediction(x, W, L)[0].shape[1]
    print(f"enc_depth {enc_d} dec_depth {dec_d} (S={S}): persistent {out[0]:.3f} ms = {B / out[0]:.1f} k chunks/s   per-step {out[1]:.3f} ms = {B / out[1]:.1f} k chunks/s")
    bc.close()
