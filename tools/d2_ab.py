"""Diagnostic: the enc3/dec2 family at the C3 shape, two decoder cells on the matrix pipe (option matrix_cell = 1, the default) against
packed FMAs (matrix_cell = 0): decode launch alone, synchronous slab, streamed slabs, digest."""
import gc, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv
B, T_r, T_e, W, L = 256, 300, 30, 5, 48
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, encoder_depth=int(os.environ.get("RV_ENC", "3")), decoder_depth=2,
                   max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22)
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
bc.set_async_depth(10)
gc.disable()
for mc in (1, 0, 1, 0):
    bc.set_option("matrix_cell", mc)
    bc.set_option("profile", 1)
    for _ in range(3): tok, sc = bc.beam_search_prediction(x, W, L)
    bc.reset_profile()
    for _ in range(10): bc.beam_search_prediction(x, W, L)
    p = bc.profile()
    bc.set_option("profile", 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): bc.beam_search_prediction(x, W, L)
    torch.cuda.synchronize(); sync = (time.perf_counter() - t0) / 20
    for _ in bc.beam_search_stream((x for _ in range(10)), W, L): pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in bc.beam_search_stream((x for _ in range(60)), W, L): pass
    torch.cuda.synchronize(); st = (time.perf_counter() - t0) / 60
    h = hashlib.sha1(tok.cpu().numpy().tobytes()).hexdigest()[:8] + f" score sum {float(sc.double().sum()):.5f}"
    print(f"matrix_cell={mc}: dec_persist {p['dec_persist'][0] / p['dec_persist'][1]:.4f} ms, synchronous {sync * 1e3:.3f} ms/slab ({B / sync / 1e3:.1f} k chunks/s), "
          f"streamed {st * 1e3:.3f} ms/slab ({B / st / 1e3:.1f} k chunks/s)  digest {h}", flush=True)
bc.close()
