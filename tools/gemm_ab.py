"""Diagnostic: launch time of the encoder input-projection GEMM (k_gemm_ws, C3 raw: M = 76,800, N = 1,024, K = 256) alone on the chip for
the library named by RAVVENT_HIP_LIB (ablation builds: make gemmvar V=n GEMMFLAGS=-DRV_WS_...; results of those are invalid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = 256, 300, 30, 5, 4
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=8)
bc.init_random_weights(seed=22)
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
for _ in range(3): bc.beam_search_prediction(x, W, L)
bc.set_option("profile", 1); bc.reset_profile()
for _ in range(20): bc.beam_search_prediction(x, W, L)
p = {k: v[0] / max(v[1], 1) for k, v in bc.profile().items()}
print(f"{os.path.basename(os.environ.get('RAVVENT_HIP_LIB', 'default')):28s} {os.environ.get('RV_NOTE', ''):44s} gemm_inproj_raw {p['gemm_inproj_raw']:.4f} ms  gemm_inproj_event {p['gemm_inproj_event']:.4f}  gemm_memory {p['gemm_memory']:.4f}")
