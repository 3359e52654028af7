"""Diagnostic: the driver's protocol (5 warm-up steps, then 20 timed steps, depth 10) with and without every slab context having run a
(tiny) slab before: what the contexts the warm-up never reaches cost inside the timed region.  usage: python tools/ctx_warm_ab.py [touch_B]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = 256, 300, 30, 5, 48
tb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
def make(n, seed):
    raw, ev, _ = rv.synthetic.make_slab(n, T_r, T_e, seed=seed)
    return (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
slabs = [make(B, k) for k in range(6)]
tiny = make(tb, 99)
for touch in (0, 1, 0, 1, 2, 2):
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
    bc.init_random_weights(seed=22)
    bc.set_async_depth(10)
    gc.collect(); gc.disable()
    if touch == 1:                          # every context runs one tiny slab
        ts = [bc.submit_beam_search(tiny, W, L) for _ in range(10)]
        for t in ts: bc.collect(t)
    if touch == 2:                          # every context runs one full slab
        ts = [bc.submit_beam_search(slabs[5], W, L) for _ in range(10)]
        for t in ts: bc.collect(t)
    for _ in bc.beam_search_stream((slabs[1 + i % 4] for i in range(5)), W, L): pass      # the driver's warm-up: 5 steps
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in bc.beam_search_stream((slabs[0] for _ in range(20)), W, L): pass
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    gc.enable()
    print(f"contexts touched before the warm-up: {('no', 'tiny slab of %d' % tb, 'full slab')[touch]}: 20 timed steps {B * 20 / dt / 1e3:.1f} k chunks/s ({dt / 20 * 1e3:.4f} ms per slab)", flush=True)
    bc.close()
