import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as td
import ravvent_basecaller_amd as rv
td.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
torch.cuda.set_device(0)
B = 64
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B)
bc.init_random_weights(seed=22)
raw, ev, _ = rv.synthetic.make_slab(B, 300, 30, seed=0)
x = (torch.from_numpy(raw).cuda(), torch.from_numpy(ev).cuda())
tok, sc = rv.dist.sharded_beam_search(bc, x[0], x[1], 5, 48)
t1, s1 = bc.beam_search_prediction(x, 5, 48)
print("sharded == local:", bool((tok.cpu() == t1.cpu()).all()), float((sc.cpu() - s1.cpu()).abs().max()), tok.device)
td.destroy_process_group()
