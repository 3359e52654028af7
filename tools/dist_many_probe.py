"""Diagnostic: where the time of dist.sharded_beam_search_many goes on ONE rank (a real one-rank RCCL group): the streamed decode of
the K slabs, the all-gather + the one host synchronisation, the read-order / extension ops -- against the plain streamed decode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29617")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import ravvent_basecaller_amd as rv
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
B, T_r, T_e, W, L, K = 256, 300, 30, 5, 48, 20
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
bc.init_random_weights(seed=22); bc.set_async_depth(10)
if os.environ.get('RV_PROFILE'): bc.set_option('profile', int(os.environ['RV_PROFILE']))
raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
import gc; gc.collect(); gc.disable()
def plain(k):
    for out in bc.beam_search_stream((x for _ in range(k)), W, L): pass
def many(k): return rv.dist.sharded_beam_search_many(bc, [x] * k, W, L, slab=B)
def stream(k):
    for out in rv.dist.sharded_beam_search_stream(bc, (x for _ in range(k)), W, L, slab=B): pass
for name, f in (("plain stream", plain), ("many (one gather)", many), ("gather per slab", stream), ("plain stream", plain), ("many (one gather)", many)):
    f(5); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(K); dist.barrier(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:20s} {dt / K * 1e3:.4f} ms per slab  ({B * K / dt:.0f} chunks/s)", flush=True)
# phases of the one-gather form
import ravvent_basecaller_amd.dist as D
orig = dist.all_gather_into_tensor
marks = {}
def ag(*a, **kw):
    torch.cuda.synchronize(); marks["pre"] = time.perf_counter(); r = orig(*a, **kw); torch.cuda.synchronize(); marks["post"] = time.perf_counter(); return r
D.dist.all_gather_into_tensor = ag
t0 = time.perf_counter(); many(K); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"decode loop {1e3 * (marks['pre'] - t0):.3f} ms, all-gather {1e3 * (marks['post'] - marks['pre']):.3f} ms, after {1e3 * (t1 - marks['post']):.3f} ms")
dist.destroy_process_group()
