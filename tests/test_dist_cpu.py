"""World-size-2 `gloo` test of the N>1 path: contiguous chunk sharding + one all-gather, with
the CPU oracle standing in for each rank's device engine.  The gathered slab must be
byte-identical to the single-process result, including when the shards stop decoding at
different steps."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OracleEngine:
    """Minimal stand-in with the Basecaller attributes dist.sharded_beam_search touches."""
    input_data_type = "joint"
    output_end_token = 1

    def __init__(self, cfg, blob):
        self.cfg, self.blob = cfg, blob

    def beam_search_prediction(self, input_data, beam_width, max_output_len):
        from oracle import cpu_port
        raw, ev = input_data
        if raw.shape[0] == 0:
            return torch.zeros((0, 0), dtype=torch.int32), torch.zeros((0, 0))
        tok, sc = cpu_port.run(self.cfg.oracle_cfg(), self.cfg.enc_depth, 7, self.blob, raw, ev,
                               beam_width, max_output_len, nthreads=2)
        return torch.from_numpy(tok.copy()), torch.from_numpy(sc.copy())


def _worker(rank, world, port, n, eos_bias, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ravvent_basecaller_amd as rv
    cfg = rv.RvConfig()
    flat = rv.weights.init_weights(cfg, seed=4)
    flat["b_fc"][1] = eos_bias                     # steer how early '^' shows up
    eng = _OracleEngine(cfg, rv.weights.pack(cfg, flat))
    raw, ev, _ = rv.synthetic.make_slab(n, 24, 6, seed=9)
    tok, sc = rv.dist.sharded_beam_search(eng, raw, ev, beam_width=3, max_output_len=9)
    if rank == 0:
        q.put((tok.numpy(), sc.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,eos_bias", [(7, 0.0), (6, 3.0)])
def test_sharded_equals_single(rv, n, eos_bias):
    from oracle import cpu_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, eos_bias, q)) for r in range(2)]
    for p in procs:
        p.start()
    tok, sc = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = rv.RvConfig()
    flat = rv.weights.init_weights(cfg, seed=4)
    flat["b_fc"][1] = eos_bias
    raw, ev, _ = rv.synthetic.make_slab(n, 24, 6, seed=9)
    rtok, rsc = cpu_port.run(cfg.oracle_cfg(), 2, 7, rv.weights.pack(cfg, flat), raw, ev, 3, 9)
    assert tok.shape == rtok.shape
    assert (tok == rtok).all()
    assert np.array_equal(sc, rsc)
