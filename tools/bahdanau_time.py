"""Diagnostic: per-slab time of the C3 shape with Bahdanau and with Luong attention (persistent decode), GC off."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gc.disable()
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = 256, 300, 30, 5, 48
for att in ("bahdanau", "luong"):
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, attention_type=att, honor_attention_type=True,
                       max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
    bc.init_random_weights(seed=22)
    raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
    x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
    bc.set_option("profile", 1)
    for _ in range(3): bc.beam_search_prediction(x, W, L)
    bc.reset_profile()
    t0 = time.perf_counter()
    for _ in range(10): tok, _ = bc.beam_search_prediction(x, W, L)
    dt = (time.perf_counter() - t0) / 10
    print(att, "ms/step", round(dt * 1e3, 3), "S", tok.shape[1], {k: round(v[0] / v[1], 4) for k, v in bc.profile().items()})
    bc.close()
