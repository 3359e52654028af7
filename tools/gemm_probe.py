"""Timing probes of the split-f16 GEMMs (memory projection, ncb = 1; encoder input projection, ncb = 4) with their stores / A loads /
MFMAs switched off (results invalid).  Needs the diagnostic build: make -C ravvent-basecaller_amd/csrc gemm_diag; run as
RAVVENT_HIP_LIB=.../libravvent_hip_gemm_diag.so RV_GEMM_DBG=<bits> python tools/gemm_probe.py   (bits: 1 no C stores, 2 no A loads, 4 no MFMAs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = 256, 300, 30, 5, 48
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22)
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
bc.set_option("wide_recurrence", 1)
for _ in range(3): bc.beam_search_prediction(x, W, L)
bc.set_option("profile", 1); bc.reset_profile()
for _ in range(8): bc.beam_search_prediction(x, W, L)
p = {k: round(v[0] / v[1], 4) for k, v in bc.profile().items() if k.startswith("gemm")}
print(f"RV_GEMM_DBG={os.environ.get('RV_GEMM_DBG', '0')}: {p}")
