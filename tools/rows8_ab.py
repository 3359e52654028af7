"""Diagnostic: the synchronous call by recurrence form -- wide_recurrence 1 (matrix pipe, 16 chunks per workgroup), 2 (8 chunks per
workgroup: the latency form), 0 (packed FMA), -1 (per-call choice) -- at several slab sizes: chunks/s, the recurrence launches alone,
and the largest score difference / share of identical token rows against form 1.  usage: python tools/rows8_ab.py [B ...]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
T_r, T_e, W, L = 300, 30, 5, 48
for B in ([int(x) for x in sys.argv[1:]] or [64, 128, 256, 512, 1024]):
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
    bc.init_random_weights(seed=22)
    raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
    x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
    ref = None
    gc.disable()
    for form in (1, 2, 0, -1, 1, 2):
        bc.set_option("wide_recurrence", form)
        for _ in range(3): tok, sc = bc.beam_search_prediction(x, W, L)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): tok, sc = bc.beam_search_prediction(x, W, L)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        bc.set_option("profile", 1); bc.reset_profile()
        for _ in range(5): bc.beam_search_prediction(x, W, L)
        p = {k: v[0] / max(v[1], 1) for k, v in bc.profile().items()}
        bc.set_option("profile", 0)
        t, s = tok.cpu().numpy(), sc.cpu().numpy()
        if ref is None: ref = (t, s)
        rec = {k: round(v, 4) for k, v in p.items() if k.startswith("lstm_rec")}
        print(f"B={B:5d} wide_recurrence={form:2d}: {dt * 1e3:.3f} ms/slab = {B / dt / 1e3:.1f} k chunks/s; {rec}; vs form 1: max |score diff| "
              f"{np.abs(s - ref[1]).max():.2e}, identical token rows {(t == ref[0]).all(axis=(1, 2)).mean() if t.ndim == 3 else (t == ref[0]).all(axis=1).mean():.4f}", flush=True)
    gc.enable()
    bc.close()
