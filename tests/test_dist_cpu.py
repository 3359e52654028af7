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


    def beam_search_call_arrays(self, input_data, beam_width, max_output_len):
        """Host form of rv_beam_search_calls (tokens_to_nuc_sequences + calc_prob_logits_beam_search_scores[:len])."""
        import ravvent_basecaller_amd as rv
        tok, sc = self.beam_search_prediction(input_data, beam_width, max_output_len)
        B, steps = input_data[0].shape[0], max(int(max_output_len) - 1, 0)
        bases = np.zeros((B, steps), np.uint8); probs = np.zeros((B, steps), np.float32); lens = np.zeros(B, np.int32)
        if B:
            seqs = rv.data_loader.tokens_to_strings(tok.numpy())
            pr = rv.utils.calc_prob_logits_beam_search_scores(sc).numpy()
            for i, sq in enumerate(seqs):
                lens[i] = len(sq)
                bases[i, :len(sq)] = np.frombuffer(sq.encode(), np.uint8)
                probs[i, :pr.shape[1]] = pr[i]
        return bases, probs, lens


    # asynchronous calls, emulated in order (the host logic of dist.sharded_beam_search_stream / Basecaller.beam_search_stream)
    async_depth = 2

    def submit_beam_search(self, input_data, beam_width, max_output_len):
        return {"result": self.beam_search_prediction(input_data, beam_width, max_output_len)}

    def collect(self, call):
        return call.pop("result")

    def beam_search_calls(self, input_data, beam_width, max_output_len):
        bases, probs, lens = self.beam_search_call_arrays(input_data, beam_width, max_output_len)
        return ([bytes(bases[i, :n]).decode() for i, n in enumerate(lens)], [probs[i, :n] for i, n in enumerate(lens)])


def _base_emitting_weights(rv, cfg, seed=4):
    flat = rv.weights.init_weights(cfg, seed=seed)
    flat["b_fc"][3:7] += 1.5          # favour a/c/g/t so that the calls are visible strings
    flat["b_fc"][cfg.end_token] -= 1.0
    return flat


def _read_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ravvent_basecaller_amd as rv
    cfg = rv.RvConfig()
    eng = _OracleEngine(cfg, rv.weights.pack(cfg, _base_emitting_weights(rv, cfg)))
    sig, lab = rv.synthetic.make_read(260, seed=5)
    res = rv.evaluator.PerformanceEvaluator(eng, fused_postprocessing=True).run_read_sharded(sig, lab, chunk_size=16, beam_width=3)
    q.put((rank, res["merged_seq"], res["chunks_num"], res["chunks_local"]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_read_equals_single_process(rv):
    """BASELINE configs 4/5 control flow on CPU: a read's chunks sharded over 2 ranks -> one all-gather -> C++ merger
    on rank 0 == the single-process evaluator on the same read (gloo; the CPU oracle stands in for each rank's GPU)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 77) % 2000
    procs = [ctx.Process(target=_read_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, merged, n, nloc = q.get(timeout=300)
        got[r] = (merged, n, nloc)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = rv.RvConfig()
    eng = _OracleEngine(cfg, rv.weights.pack(cfg, _base_emitting_weights(rv, cfg)))
    sig, lab = rv.synthetic.make_read(260, seed=5)
    single = rv.evaluator.PerformanceEvaluator(eng, fused_postprocessing=True).run_read(sig, lab, chunk_size=16, beam_width=3)
    assert got[1][0] is None and got[0][1] == single["chunks_num"] and got[0][2] + got[1][2] == single["chunks_num"]
    assert len(single["merged_seq"]) > 50
    assert got[0][0] == single["merged_seq"]


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
    tok, sc = rv.dist.sharded_beam_search(eng, raw, ev, beam_width=3, max_output_len=9, slab=2 if n > 6 else None)
    if rank == 0:
        q.put((tok.numpy(), sc.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,eos_bias", [(7, 0.0), (6, 3.0), (1, 0.0), (9, 2.0)])
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


def _stream_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ravvent_basecaller_amd as rv
    cfg = rv.RvConfig()
    flat = rv.weights.init_weights(cfg, seed=4)
    flat["b_fc"][1] = 1.5
    eng = _OracleEngine(cfg, rv.weights.pack(cfg, flat))
    slabs = [rv.synthetic.make_slab(n, 24, 6, seed=20 + n)[:2] for n in (5, 1, 8, 4, 7)]      # n = 1: rank 1's shard is empty
    outs = list(rv.dist.sharded_beam_search_stream(eng, slabs, beam_width=3, max_output_len=9, slab=3))   # (8 chunks: shard of 4 > slab 3)
    if rank == 0:
        q.put([(t.numpy(), s.numpy()) for t, s in outs])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_stream_equals_single(rv):
    """dist.sharded_beam_search_stream: slab k + 1's shard is submitted before slab k is collected and gathered; one collective per
    slab, results in slab order, identical to the single-process decode -- with an empty shard and a shard larger than the slab limit."""
    from oracle import cpu_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 555) % 2000
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = rv.RvConfig()
    flat = rv.weights.init_weights(cfg, seed=4)
    flat["b_fc"][1] = 1.5
    blob = rv.weights.pack(cfg, flat)
    assert len(outs) == 5
    for (tok, sc), n in zip(outs, (5, 1, 8, 4, 7)):
        raw, ev, _ = rv.synthetic.make_slab(n, 24, 6, seed=20 + n)
        rtok, rsc = cpu_port.run(cfg.oracle_cfg(), 2, 7, blob, raw, ev, 3, 9)
        assert tok.shape == rtok.shape and (tok == rtok).all() and np.array_equal(sc, rsc), n


def _many_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ravvent_basecaller_amd as rv
    cfg = rv.RvConfig()
    flat = rv.weights.init_weights(cfg, seed=4)
    flat["b_fc"][1] = 1.5
    eng = _OracleEngine(cfg, rv.weights.pack(cfg, flat))
    slabs = [rv.synthetic.make_slab(n, 24, 6, seed=20 + n)[:2] for n in (5, 1, 8, 4, 7)]      # n = 1: rank 1's shard is empty
    outs = rv.dist.sharded_beam_search_many(eng, slabs, beam_width=3, max_output_len=9, slab=3)   # (8 chunks: shard of 4 > slab 3)
    if rank == 0:
        q.put([(t.numpy(), s.numpy()) for t, s in outs])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_many_one_gather_equals_single(rv):
    """dist.sharded_beam_search_many: every rank decodes its shards of ALL the slabs, ONE all-gather at the end; per slab identical to
    the single-process decode -- with an empty shard, ragged slab sizes and a shard larger than the slab limit."""
    from oracle import cpu_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 777) % 2000
    procs = [ctx.Process(target=_many_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = rv.RvConfig()
    flat = rv.weights.init_weights(cfg, seed=4)
    flat["b_fc"][1] = 1.5
    blob = rv.weights.pack(cfg, flat)
    assert len(outs) == 5
    for (tok, sc), n in zip(outs, (5, 1, 8, 4, 7)):
        raw, ev, _ = rv.synthetic.make_slab(n, 24, 6, seed=20 + n)
        rtok, rsc = cpu_port.run(cfg.oracle_cfg(), 2, 7, blob, raw, ev, 3, 9)
        assert tok.shape == rtok.shape and (tok == rtok).all() and np.array_equal(sc, rsc), n
