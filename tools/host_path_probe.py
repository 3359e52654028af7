"""Diagnostic: host-buffer vs device-resident call time of beam_search_prediction (C3 slab)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=256, max_raw_len=300, max_event_len=30, max_output_len=48)
bc.init_random_weights(22)
raw, ev, _ = rv.synthetic.make_slab(256, 300, 30, seed=0)
draw, dev_ = torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda()
def timeit(inp, n=30):
    for _ in range(5): bc.beam_search_prediction(inp, 5, 48)
    t = time.perf_counter()
    for _ in range(n): bc.beam_search_prediction(inp, 5, 48)
    return (time.perf_counter() - t) / n * 1e3
print("device-resident ms/slab", timeit((draw, dev_)))
print("host-buffer     ms/slab", timeit((raw, ev)))
print("device-resident ms/slab", timeit((draw, dev_)))
bc.set_option("profile", 1)
bc.reset_profile(); timeit((raw, ev), 10); p = bc.profile()
print("host path device ms:", {k: round(v[0] / v[1], 3) for k, v in p.items()})
