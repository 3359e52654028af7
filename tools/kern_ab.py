"""Diagnostic: per-kernel launch durations of one C3 slab (synchronous calls, every launch alone on the chip; matrix-pipe encoder form)
and the streamed rate, for the library named by RAVVENT_HIP_LIB -- A/B of kernel variants."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = 256, 300, 30, 5, 48
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22)
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
bc.set_option("wide_recurrence", 1)
for _ in range(3): tok, sc = bc.beam_search_prediction(x, W, L)
bc.set_option("profile", 1); bc.reset_profile()
for _ in range(10): bc.beam_search_prediction(x, W, L)
p = {k: v[0] / max(v[1], 1) for k, v in bc.profile().items()}; bc.set_option("profile", 0)
gc.disable(); bc.set_async_depth(10)
for _ in bc.beam_search_stream((x for _ in range(30)), W, L): pass
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in bc.beam_search_stream((x for _ in range(100)), W, L): pass
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
import hashlib
print(os.path.basename(os.environ.get("RAVVENT_HIP_LIB", "default")), " ".join(f"{k}={v:.4f}" for k, v in sorted(p.items()) if v > 0.004),
      f"| streamed {dt * 1e3:.4f} ms/slab = {B / dt:.0f} chunks/s | digest", hashlib.sha1(tok.cpu().numpy().tobytes() + sc.cpu().numpy().tobytes()).hexdigest()[:10])
