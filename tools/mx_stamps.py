"""Diagnostic: where a step of the matrix-pipe recurrence (k_lstm_rec_mx) goes -- cycle sums of workgroup (0, 0), wave 0 over all steps.
RV_REC_STAMPS=1: raw layer 0; =2: raw layer 1.  Needs the stamps build of the kernel:
  make -C ravvent-basecaller_amd/csrc mxvar V=1 MXFLAGS=-DRV_MX_STAMPS
  RAVVENT_HIP_LIB=$PWD/ravvent-basecaller_amd/csrc/libravvent_hip_m1.so RV_REC_STAMPS=1 python tools/mx_stamps.py"""
import os, sys
os.environ.setdefault("RV_REC_STAMPS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = 256, 300, 30, 5, 4
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=8)
bc.init_random_weights(seed=22)
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
for _ in range(3):
    bc.beam_search_prediction(x, W, L)
    ts = bc.get_tensor("rec_stamps")
    T = max(ts[4], 1)
    print(f"layer {int(os.environ['RV_REC_STAMPS']) - 1}: cycles per step: LDS reads + MFMAs + gate sums {ts[0] / T:.0f}, cell update {ts[1] / T:.0f}, image + store issue {ts[2] / T:.0f}, barrier {ts[3] / T:.0f}; step {sum(ts[:4]) / T:.0f} ({int(T)} steps)")
