"""Diagnostic: throughput of K Basecaller handles driven by K host threads at once (each handle owns its stream; ctypes calls drop
the GIL), against one handle: do slabs of different handles fill each other's idle CUs (decode chunks that finish early, launch
gaps)?  usage: python tools/concurrent_handles.py [B] [T_r] [L]"""
import gc, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T_r = int(sys.argv[2]) if len(sys.argv) > 2 else 300
L = int(sys.argv[3]) if len(sys.argv) > 3 else 48
T_e, W, N = 30, 5, 40
gc.disable()
def mk(seed):
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
    bc.init_random_weights(seed=22)
    bc.reuse_output_buffers = True
    raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=seed)
    return bc, (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
# warm the GPU up first (clocks ramp over the first tens of milliseconds of load)
_w, _x = mk(99)
for _ in range(60): _w.beam_search_prediction(_x, W, L)
_w.close()
for K in (1, 2, 3):
    hs = [mk(i) for i in range(K)]
    for bc, x in hs:
        for _ in range(60): bc.beam_search_prediction(x, W, L)   # (re-)warm: creating the handles left the GPU idle long enough to clock down
    torch.cuda.synchronize()
    def work(bc, x, n):
        for _ in range(n): bc.beam_search_prediction(x, W, L)
    th = [threading.Thread(target=work, args=(bc, x, N)) for bc, x in hs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"{K} handle(s): {K * N * B / dt:9.0f} chunks/s  ({dt / N * 1e3:.3f} ms per round of {K} slab(s))")
    for bc, _ in hs: bc.close()
