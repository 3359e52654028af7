"""Diagnostic: per-wave cycle sums of the raw layer-0 recurrence (workgroup (0,0)): cycles between barriers (busy) and cycles inside
barriers (wait), per step.  Waves 0-7 = packed-FMA waves, 8-11 = cell-update waves.  Needs the -DRV_REC_STAMPS build:
  make -C ravvent-basecaller_amd/csrc stamps && RAVVENT_HIP_LIB=ravvent-basecaller_amd/csrc/libravvent_hip_stamps.so python tools/rec_stamps.py [B]"""
import os, sys
os.environ.setdefault("RV_REC_STAMPS", "1")      # 2 = the fused layer-1 kernel (waves 0-7 recurrence, 8-11 projection)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T_r, T_e, W, L = 300, 30, 5, 48
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22)
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
for _ in range(5):
    bc.beam_search_prediction(x, W, L)
ts = bc.get_tensor("rec_stamps").reshape(12, 2)
if len(sys.argv) > 2: bc.set_option("split_projection", int(sys.argv[2]))
for _ in range(2):
    bc.beam_search_prediction(x, W, L)
ts = bc.get_tensor("rec_stamps").reshape(12, 2)
print(f"B={B} T={T_r}: per wave (busy, barrier wait) cycles per step:")
for w in range(12):
    print(f"  wave {w:2d} ({'fma ' if w < 8 else 'tail'}): busy {ts[w,0]/T_r:8.1f}   wait {ts[w,1]/T_r:8.1f}   sum {ts[w].sum()/T_r:8.1f}")
