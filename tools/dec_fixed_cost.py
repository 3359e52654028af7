"""Diagnostic: the decode launch by number of decode steps (max_output_len) at the C3 shape -- what the launch costs before / outside its
step loop (memory -> resident fragments, weight fragments -> LDS, launch and drain)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W = 256, 300, 30, 5
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
for L in (2, 3, 5, 9, 17, 48):
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
    bc.init_random_weights(seed=22)
    for _ in range(3): bc.beam_search_prediction(x, W, L)
    bc.set_option("profile", 1); bc.reset_profile()
    for _ in range(20): bc.beam_search_prediction(x, W, L)
    p = {k: v[0] / max(v[1], 1) for k, v in bc.profile().items()}
    print(f"L={L:3d} ({L - 1} steps at most): dec_persist {p['dec_persist']:.4f} ms, dec_finalize {p['dec_finalize']:.4f}, gemm_memory {p['gemm_memory']:.4f}", flush=True)
    bc.close()
