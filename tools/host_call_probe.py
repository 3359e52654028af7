"""Diagnostic: per-call wall time of the host-buffer entry points (numpy in / numpy out), looking for stalls."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, L = 256, 200, 30, 32
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=1024, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22)
raw, ev, nuc = rv.synthetic.make_slab(B, T_r, T_e, seed=0, L=L)
for gcmode in ("gc on", "gc off"):
    if gcmode == "gc off":
        gc.collect(); gc.disable()
    for name, fn in (("prediction(host)", lambda: bc.beam_search_prediction((raw, ev), 5, L)),
                     ("call_arrays(host)", lambda: bc.beam_search_call_arrays((raw, ev), 5, L))):
        fn()
        ts = []
        for _ in range(300):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        ts = np.array(ts) * 1e3
        print(f"{gcmode:7s} {name:18s} median {np.median(ts):.2f} ms  p99 {np.percentile(ts, 99):.2f}  max {ts.max():.2f}  "
              f"calls > 5 ms: {(ts > 5).sum()}  total {ts.sum():.0f} ms")
