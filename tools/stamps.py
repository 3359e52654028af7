"""Diagnostic: in-kernel phase stamps of the attend kernel (block 0, decode step 3). RV_DBG_STAMPS=1."""
import os, sys
os.environ["RV_DBG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ravvent_basecaller_amd as rv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T_R = int(os.environ.get("RV_T_R", "300"))
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_R, max_event_len=30, max_output_len=48,
                   attention_type=os.environ.get("RV_ATT", "luong"), honor_attention_type=True)
bc.init_random_weights(seed=22)
raw, ev, _ = rv.synthetic.make_slab(B, T_R, 30, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
for _ in range(3):
    bc.beam_search_prediction(x, 5, 48)
    ts = bc.get_tensor("dbg_stamps")
    if os.environ.get("RV_PERSIST", "1") != "0":
        pn = [("gates", 0, 2), ("att-h", 2, 3), ("scores", 3, 4), ("softmax", 4, 5), ("context", 5, 6), ("merge", 6, 8),
              ("logits", 8, 9), ("beam||cell-GEMV", 9, 10)]
        print(" ".join(f"{n}={ts[j]-ts[i]:.0f}" for n, i, j in pn), "step total", ts[10] - ts[0],
              f"| softmax: local {ts[14]-ts[4]:.0f} + barrier {ts[15]-ts[14]:.0f} + merge and image {ts[5]-ts[15]:.0f} | beam step (wave 0) {ts[11]-ts[9]:.0f}, wave 1's cell product {ts[13]-ts[12]:.0f} (starts {ts[12]-ts[9]:.0f} after the logits barrier)")
        continue
    names = ["entry", "prologue", "qprime", "sweep", "merge", "E att", "F logits", "G beam", "H/end"]
    print(" ".join(f"{n}={ts[i]-ts[i-1] if i else 0:.0f}" for i, n in enumerate(names[:8])), "total", ts[7])
