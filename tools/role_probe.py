"""Diagnostic: time of the fused recurrence+projection kernel with one role's math disabled (RV_DBG_ROLE)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# RV_DBG_ROLE is compiled only into the diagnostic library (`make -C ravvent-basecaller_amd/csrc diag`): the product library ignores it
os.environ.setdefault("RAVVENT_HIP_LIB", os.path.join(ROOT, "ravvent-basecaller_amd", "csrc", "libravvent_hip_diag.so"))
if not os.path.exists(os.environ["RAVVENT_HIP_LIB"]):
    raise SystemExit(f"role_probe.py: {os.environ['RAVVENT_HIP_LIB']} not found -- build it with: make -C ravvent-basecaller_amd/csrc diag")
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = 256, 300, 30, 5, 8
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=48)
bc.init_random_weights(seed=22)
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
bc.set_option("profile", 1)
for _ in range(5): bc.beam_search_prediction(x, W, L)
bc.reset_profile()
for _ in range(10): bc.beam_search_prediction(x, W, L)
print("RV_DBG_ROLE", os.environ.get("RV_DBG_ROLE", "0"), {k: round(v[0] / v[1], 4) for k, v in bc.profile().items() if "rec" in k})
