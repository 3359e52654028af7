"""Diagnostic: throughput of back-to-back slabs through the asynchronous calls (rv_beam_search_submit_dev / collect_dev) against
the synchronous call, for several depths and both recurrence forms.  Usage: async_time.py [B,T_r,T_e,W,L] [n_slabs] [depths, e.g. 2,4,8] [wide_recurrence values, e.g. 1]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = (int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (256, 300, 30, 5, 48)
n_slabs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
depths = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 4, 6, 8]
wides = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [-1, 0, 1]
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22)
slabs = []
for i in range(4):
    raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=i)
    slabs.append((torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda()))
gc.disable()
ref = [bc.beam_search_prediction(x, W, L) for x in slabs]
ref = [(t.cpu().numpy().copy(), s.cpu().numpy().copy()) for t, s in ref]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n_slabs):
    bc.beam_search_prediction(slabs[i % 4], W, L)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"B={B} T=({T_r},{T_e}) W={W} L={L}: synchronous {dt / n_slabs * 1e3:.3f} ms/slab -> {B * n_slabs / dt / 1e3:.1f} k chunks/s", flush=True)
for wide in wides:
    bc.set_option("wide_recurrence", wide)
    for depth in depths:
        bc.set_async_depth(depth)
        outs = list(bc.beam_search_stream([slabs[i % 4] for i in range(8)], W, L))      # warm-up: creates the contexts
        ok = all((o[0].cpu().numpy() == ref[i % 4][0]).all() and np.abs(o[1].cpu().numpy() - ref[i % 4][1]).max() < 1e-4 for i, o in enumerate(outs))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in bc.beam_search_stream((slabs[i % 4] for i in range(n_slabs)), W, L):
            pass
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"  wide={wide:2d} depth={depth}: {dt / n_slabs * 1e3:.3f} ms/slab -> {B * n_slabs / dt / 1e3:.1f} k chunks/s   results match sync: {ok}", flush=True)
bc.close()
