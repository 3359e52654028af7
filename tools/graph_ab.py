"""Diagnostic: option slab_graph (one hipGraphLaunch per slab) against ten kernel launches per slab -- host time of a submit,
streamed throughput at the driver's settings (20 steps, depth 10, from an idle GPU), settled throughput, the synchronous call, and a
byte-for-byte comparison of the results.  usage: python tools/graph_ab.py [B T_r T_e L]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ravvent_basecaller_amd as rv

B, T_r, T_e, L = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (256, 300, 30, 48)
W, depth = 5, 10
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22)
slabs = []
for k in range(4):
    raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=k)
    slabs.append((torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda()))
bc.set_async_depth(depth)
res = {}
for graph in (0, 1, 0, 1):
    bc.set_option("slab_graph", graph)
    outs = [(t.cpu().numpy().copy(), s.cpu().numpy().copy()) for t, s in bc.beam_search_stream(slabs * 3, W, L)]     # warm-up, capture
    res.setdefault(graph, outs)
    gc.collect(); gc.disable()
    # host time of a submit (queue never full: collect right after the burst)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        calls = [bc.submit_beam_search(slabs[i % 4], W, L) for i in range(depth)]
        ts.append((time.perf_counter() - t0) / depth)
        for c in calls:
            bc.collect(c)
    # the driver's settings: 20 steps from an idle GPU
    rates = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in bc.beam_search_stream((slabs[0] for _ in range(20)), W, L):
            pass
        torch.cuda.synchronize()
        rates.append(B * 20 / (time.perf_counter() - t0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in bc.beam_search_stream((slabs[0] for _ in range(200)), W, L):
        pass
    torch.cuda.synchronize()
    settled = B * 200 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(20):
        bc.beam_search_prediction(slabs[0], W, L)
    torch.cuda.synchronize()
    sync = B * 20 / (time.perf_counter() - t0)
    gc.enable()
    print(f"slab_graph={graph}: submit {np.median(ts) * 1e6:.1f} us (min {min(ts) * 1e6:.1f}); 20 steps from idle {np.median(rates) / 1e3:.1f} k chunks/s "
          f"(max {max(rates) / 1e3:.1f}); settled {settled / 1e3:.1f} k; synchronous {sync / 1e3:.1f} k", flush=True)
same = all((a[0] == b[0]).all() and np.array_equal(a[1], b[1]) for a, b in zip(res[0], res[1]))
print("results identical:", same)
bc.close()
