"""Diagnostic: duration of the persistent decode launch against the number of chunks (= CUs streaming the cell weights from L2 at
once).  Round 3: 0.434 ms at 16 chunks, 0.477 ms at 256 -- the cell product's weight stream runs at ~35 B/clk per CU whether 16 or 256
CUs stream, i.e. it is bound by the requests a CU keeps in flight, not by the L2's all-CU ceiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
T_r, T_e, W, L = 300, 30, 5, 48
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=512, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22)
for B in (16, 32, 64, 128, 192, 256, 384, 512):
    raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
    x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
    for _ in range(3): bc.beam_search_prediction(x, W, L)
    bc.set_option("profile", 1); bc.reset_profile()
    for _ in range(8): bc.beam_search_prediction(x, W, L)
    p = bc.profile(); bc.set_option("profile", 0)
    cs = bc.get_tensor("chunk_steps")
    print(f"B={B:4d}: dec_persist {p['dec_persist'][0] / p['dec_persist'][1]:.4f} ms  max steps {int(cs.max())} mean {cs.mean():.1f}", flush=True)
