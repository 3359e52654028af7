"""Diagnostic: wall time per C3 slab (device-resident inputs) with the library's profiling events off / on."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = 256, 300, 30, 5, 48
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22)
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
gc.disable()
for prof in (0, 1, 0, 1):
    bc.set_option("profile", prof)
    for _ in range(5):
        bc.beam_search_prediction(x, W, L)
    torch.cuda.synchronize()
    ts = []
    for _ in range(60):
        t = time.perf_counter(); bc.beam_search_prediction(x, W, L); ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e3
    print(f"profile={prof}: mean {ts.mean():.3f} ms  median {np.median(ts):.3f}  min {ts.min():.3f}  max {ts.max():.3f}  >2.8ms: {(ts > 2.8).sum()}")
bc.set_option("profile", 0)
for side in (0, 1, 0, 1):
    bc.set_option("concurrent_encoders", side)
    for _ in range(5):
        bc.beam_search_prediction(x, W, L)
    ts = []
    for _ in range(60):
        t = time.perf_counter(); bc.beam_search_prediction(x, W, L); ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e3
    print(f"concurrent_encoders={side}: mean {ts.mean():.3f} ms  median {np.median(ts):.3f}  min {ts.min():.3f}")
bc.set_option("fused_projection", 0)
ref = bc.beam_search_prediction(x, W, L)
ref_enc = bc.get_tensor("enc_output")
for fuse in (0, 1, 0, 1):
    bc.set_option("fused_projection", fuse)
    for _ in range(5):
        out = bc.beam_search_prediction(x, W, L)
    enc = bc.get_tensor("enc_output")
    print("   max |enc - enc_unfused|", float(np.abs(enc - ref_enc).max()), "tokens equal", bool((out[0] == ref[0]).all()))
    ts = []
    for _ in range(60):
        t = time.perf_counter(); bc.beam_search_prediction(x, W, L); ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e3
    print(f"fused_projection={fuse}: mean {ts.mean():.3f} ms  median {np.median(ts):.3f}  min {ts.min():.3f}")
bc.set_option("profile", 1); bc.reset_profile()
for _ in range(10): bc.beam_search_prediction(x, W, L)
print({k: round(v[0] / v[1], 4) for k, v in bc.profile().items()})
