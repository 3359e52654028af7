"""Diagnostic: per-slab time of greedy_search_prediction vs beam_search_prediction (beam 1 / 5) at the C3 shape."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, L = 256, 300, 30, 48
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22)
gc.disable()
for name, fn in (("greedy", lambda: bc.greedy_search_prediction(x, L)), ("beam 1", lambda: bc.beam_search_prediction(x, 1, L)),
                 ("beam 5", lambda: bc.beam_search_prediction(x, 5, L))):
    for _ in range(8): fn()
    ts = []
    for _ in range(30):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    print(f"{name}: median {np.median(ts)*1e3:.3f} ms  S={fn()[0].shape[1]}")
