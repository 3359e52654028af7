"""Diagnostic: the driver's timed region (20 C3 steps from an idle GPU, depth 10) and the settled stream, several repetitions."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = 256, 300, 30, 5, 48
depth = int(os.environ.get("RV_DEPTH", "10"))
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22)
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
bc.set_async_depth(depth)
gc.disable()
for _ in bc.beam_search_stream((x for _ in range(5)), W, L): pass
rates = []
for rep in range(6):
    torch.cuda.synchronize(); time.sleep(0.05)
    for _ in bc.beam_search_stream((x for _ in range(5)), W, L): pass       # the driver's 5 warm-up steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in bc.beam_search_stream((x for _ in range(20)), W, L): pass
    torch.cuda.synchronize()
    rates.append(B * 20 / (time.perf_counter() - t0))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in bc.beam_search_stream((x for _ in range(200)), W, L): pass
torch.cuda.synchronize()
settled = B * 200 / (time.perf_counter() - t0)
print(f"depth {depth} prio {os.environ.get('RV_CTX_PRIO', '0')}: 20 steps after 5 warm-up steps {np.median(rates) / 1e3:.1f} k chunks/s (min {min(rates) / 1e3:.1f}, max {max(rates) / 1e3:.1f}); settled {settled / 1e3:.1f} k")
