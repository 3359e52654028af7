"""Diagnostic: per-kernel time of one slab with the packed-FMA recurrence (wide_recurrence=0) and with the matrix-pipe recurrence
(wide_recurrence=1: 16 chunks per workgroup, lstm_mx.hip), at the C3 and R shapes and at a few slab sizes."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
shapes = [(256, 300, 30, 5, 48), (1024, 200, 30, 5, 32), (512, 300, 30, 5, 48), (1024, 300, 30, 5, 48)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for B, T_r, T_e, W, L in shapes:
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
    bc.init_random_weights(seed=22)
    raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
    x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
    gc.disable()
    res = {}
    for wide in (0, 1):
        bc.set_option("wide_recurrence", wide)
        bc.set_option("profile", 0)
        for _ in range(4):
            out = bc.beam_search_prediction(x, W, L)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            t = time.perf_counter(); bc.beam_search_prediction(x, W, L); ts.append(time.perf_counter() - t)
        ts = np.array(ts) * 1e3
        res[wide] = (out[0].cpu().numpy().copy(), out[1].cpu().numpy().copy(), bc.get_tensor("enc_output"))
        bc.set_option("profile", 1); bc.reset_profile()
        for _ in range(6): bc.beam_search_prediction(x, W, L)
        prof = {k: round(v[0] / v[1], 4) for k, v in bc.profile().items()}
        print(f"B={B} T=({T_r},{T_e}) W={W} L={L} wide={wide}: median {np.median(ts):.3f} ms  min {ts.min():.3f}  -> {B / np.median(ts):.1f} k chunks/s   {prof}", flush=True)
    print("   max |enc_wide - enc_fma|", float(np.abs(res[1][2] - res[0][2]).max()), " tokens equal rows", int((res[1][0] == res[0][0]).all(axis=1).sum()), "of", B,
          " max |score diff|", float(np.abs(res[1][1] - res[0][1]).max()), flush=True)
    gc.enable()
    bc.close()
