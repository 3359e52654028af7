"""Diagnostic: duration of the persistent decode launch (alone on the chip, synchronous calls) at the C3 and R shapes for the library
named by RAVVENT_HIP_LIB, with a digest of the results so that variants of one kernel can be compared (tools/dec_ab.sh)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
for name, B, T_r, T_e, W, L in (("C3", 256, 300, 30, 5, 48), ("R", 1024, 200, 30, 5, 34)):
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
    flat = bc.init_random_weights(seed=22)
    if "--emitting" in sys.argv:
        bc.set_weights_flat(rv.weights.base_calling_weights(bc.cfg, seed=22))
    for kv in os.environ.get("RV_OPTS", "").split(","):
        if "=" in kv: bc.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
    x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
    for _ in range(3): tok, sc = bc.beam_search_prediction(x, W, L)
    bc.set_option("profile", 1); bc.reset_profile()
    for _ in range(10): bc.beam_search_prediction(x, W, L)
    p = bc.profile(); bc.set_option("profile", 0)
    cs = bc.get_tensor("chunk_steps")
    h = hashlib.sha1(tok.cpu().numpy().tobytes()).hexdigest()[:8] + " " + hashlib.sha1(sc.cpu().numpy().tobytes()).hexdigest()[:8] + f" score sum {float(sc.double().sum()):.6f}"
    print(f"{os.path.basename(os.environ.get('RAVVENT_HIP_LIB', 'default')) + ' ' + os.environ.get('RV_OPTS', ''):36s} {name}: dec_persist {p['dec_persist'][0] / p['dec_persist'][1]:.4f} ms  "
          f"mean steps {cs.mean():.1f}  digest {h}", flush=True)
    bc.close()
