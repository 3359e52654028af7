"""BASELINE config 5 at its stand-in size (a 50 k-base synthetic read for the Lambda long reads), the product
paths that no earlier test executed on a GPU (`Basecaller.load_weights(checkpoint_prefix)`, the RCCL gather), and an
adversarial case for the split-f16 operands of the default path.

Reference chain of the read-level tests: /root/reference/ravvent_performance_evaluator.py:24-87 (slabs of 1,024 chunks ->
beam_search_prediction(beam 5) -> strings + per-base probabilities -> merger), merge :73-75, /root/reference/merger.py:155-248."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.abspath(__file__)
TOL = 1e-4
N_BASES = 50000


def _long_read(rv, n_bases=N_BASES, seed=0):
    sig, lab = rv.synthetic.make_read(n_bases, seed=seed)
    la = np.asarray(lab)
    raw, ev, nuc = rv.data_loader.snippets_to_slab(*rv.data_loader.prepare_snippets(sig, la[:, :2].astype(int), la[:, 2], 6))
    return sig, lab, raw, ev, nuc


def _read_basecaller(rv, L, max_batch=1024, device=None):
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=max_batch, max_raw_len=200,
                       max_event_len=30, max_output_len=L, device=device)
    bc.set_weights_flat(rv.weights.base_calling_weights(bc.cfg))      # calls of L - 1 letters: the merger appends (merger.cpp: n > overlap)
    return bc


def test_config5_long_read_sharded_eight_ways(rv):
    """One 50 k-base read (7.7 k chunks of <= 200 samples + <= 30 events, stride 6) through the evaluator call sequence in slabs
    of 1,024, and through the 8-GPU form of BASELINE config 5 with the eight shards decoded one after the other in this process:
    shard ranges -> `decode_range` -> the all-gather wire format laid out rank-major -> read order -> C++ merger.  Every form
    must give the same read, and the read must really have been stitched: every call is longer than the 25-letter overlap and
    the merged read is longer than half the reference."""
    from oracle import merger_oracle
    sig, lab, raw, ev, nuc = _long_read(rv)
    n, L = raw.shape[0], nuc.shape[1]
    assert n > 7000 and L > 27
    bc = _read_basecaller(rv, L)
    e = rv.evaluator.PerformanceEvaluator(bc, fused_postprocessing=True)
    whole = e.run_read(sig, lab, chunk_size=1024)
    lens = np.array([len(s) for s, _ in whole["nuc_preds"]])
    assert whole["chunks_num"] == n and whole["bases_num"] == N_BASES
    assert lens.min() > 25, lens.min()
    merged = whole["merged_seq"]
    assert len(merged) > 0.5 * N_BASES and set(merged) == set("ACGT"), len(merged)
    assert len({s for s, _ in whole["nuc_preds"]}) > 0.9 * n             # the calls differ from chunk to chunk
    # the C++ merger against the line-by-line restatement of merger.py on these 7.7 k real calls
    assert merger_oracle.merge(whole["nuc_preds"])[0] == merged
    # host post-processing (the reference's own sequence: tokens -> strings, scores -> probabilities) and the pipelined merge
    host = rv.evaluator.PerformanceEvaluator(bc, fused_postprocessing=False).run_slabs(raw, ev, nuc, bases_num=N_BASES, chunk_size=1024)
    assert [s for s, _ in host["nuc_preds"]] == [s for s, _ in whole["nuc_preds"]] and host["merged_seq"] == merged
    piped = rv.evaluator.PerformanceEvaluator(bc, pipelined_merge=True).run_slabs(raw, ev, nuc, bases_num=N_BASES, chunk_size=1024)
    assert piped["merged_seq"] == merged
    for world in (8, 5):
        rows = []
        for rank in range(world):
            lo, hi = rv.dist.shard_range(n, rank, world)
            b, p, l = e.decode_range(raw, ev, lo, hi, L, chunk_size=1024)
            rows.append(rv.dist.pack_call_arrays(b, p, l, n, L - 1, world))
        gb, gp, gl = rv.dist.unpack_call_arrays(np.concatenate(rows, axis=0), n, L - 1, world)
        assert gb.shape == (n, L - 1) and (gl == lens).all()
        assert e.merger.merge_arrays(gb, gp, gl)[0] == merged, world
    bc.close()


def _read_rank(rank, world, port, backend, q):
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import ravvent_basecaller_amd as rv
    sig, lab, raw, ev, nuc = _long_read(rv)
    L = nuc.shape[1]
    bc = _read_basecaller(rv, L, device=0)
    out = {"backend": dist.get_backend()}
    res = rv.evaluator.PerformanceEvaluator(bc, fused_postprocessing=True).run_read_sharded(sig, lab, chunk_size=1024)
    out["read"] = res["merged_seq"]
    out["chunks_local"] = res["chunks_local"]
    # the slab-level collective (tokens | score bits | steps): host inputs, then device inputs with slab-split shards
    n = 700
    t, s = rv.dist.sharded_beam_search(bc, raw[:n], ev[:n], 5, L)
    out["host"] = (t.cpu().numpy(), s.cpu().numpy(), str(t.device))
    t, s = rv.dist.sharded_beam_search(bc, torch.from_numpy(raw[:n]).cuda(), torch.from_numpy(ev[:n]).cuda(), 5, L, slab=256)
    out["dev"] = (t.cpu().numpy(), s.cpu().numpy(), str(t.device))
    # a queue of slabs with ONE collective at the end (what bench.py --gpus N times): the first of three slabs is the one above
    d_raw, d_ev = torch.from_numpy(raw[:n]).cuda(), torch.from_numpy(ev[:n]).cuda()
    many = rv.dist.sharded_beam_search_many(bc, [(d_raw, d_ev), (d_raw[:300], d_ev[:300]), (d_raw[5:6], d_ev[5:6])], 5, L, slab=256)
    out["many"] = [(t.cpu().numpy(), s.cpu().numpy(), str(t.device)) for t, s in many]
    if rank == 0:
        q.put(out)
    dist.barrier()
    bc.close()
    dist.destroy_process_group()


def _run_ranks(world, backend):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + (977 if backend == "nccl" else 1301)) % 2000
    procs = [ctx.Process(target=_read_rank, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    out, t0 = None, time.time()
    while out is None:                                   # fail fast when a rank dies before it reports
        try:
            out = q.get(timeout=5)
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f"rank process exited with {dead}"
            assert time.time() - t0 < 900, "ranks did not report within 900 s"
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    return out


def _single_process_reference(rv):
    sig, lab, raw, ev, nuc = _long_read(rv)
    L = nuc.shape[1]
    bc = _read_basecaller(rv, L)
    single = rv.evaluator.PerformanceEvaluator(bc, fused_postprocessing=True).run_read(sig, lab, chunk_size=1024)
    t, s = bc.beam_search_prediction((raw[:700], ev[:700]), 5, L)
    bc.close()
    return single["merged_seq"], t.numpy(), s.numpy()


def _check_many(rv, out, t, s):
    """dist.sharded_beam_search_many (one gather for three slabs) against single-process decodes of the same slabs."""
    sig, lab, raw, ev, nuc = _long_read(rv)
    L = nuc.shape[1]
    bc = _read_basecaller(rv, L)
    refs = [(t, s)]
    for sl in (slice(0, 300), slice(5, 6)):
        tt, ss = bc.beam_search_prediction((raw[:700][sl], ev[:700][sl]), 5, L)
        refs.append((tt.numpy(), ss.numpy()))
    bc.close()
    assert len(out["many"]) == 3
    for (mt, ms, _), (rt, rs) in zip(out["many"], refs):
        assert mt.shape == rt.shape and (mt == rt).all() and np.array_equal(ms, rs)


def test_config5_long_read_two_real_ranks(rv):
    """The shipped multi-GPU path on the 50 k-base read with two real ranks that share this box's GPU (gloo: RCCL refuses two
    ranks on one device): run_read_sharded -> one all-gather -> merger on rank 0 == the single-process read."""
    out = _run_ranks(2, "gloo")
    merged, t, s = _single_process_reference(rv)
    assert len(merged) > 0.5 * N_BASES and out["read"] == merged
    assert 3800 < out["chunks_local"] < 3900
    for key in ("host", "dev"):
        bad = np.nonzero((out[key][0] != t).any(axis=1) | (out[key][1] != s).any(axis=1))[0] if out[key][0].shape == t.shape else None
        assert out[key][0].shape == t.shape and bad.size == 0, (key, out[key][0].shape, t.shape, None if bad is None else (bad.size, bad[:12]))
    _check_many(rv, out, t, s)


def test_rccl_gather_world_one(rv):
    """The RCCL code path of the gather on the one GPU of this box: a real `nccl` process group of ONE rank, gather buffers in
    device memory, `all_gather_into_tensor` on them, the int32 views of the score bits -- through `dist.sharded_beam_search`
    (host and device inputs) and `evaluator.run_read_sharded` == the single-process results."""
    out = _run_ranks(1, "nccl")
    assert out["backend"] == "nccl"
    merged, t, s = _single_process_reference(rv)
    assert out["read"] == merged and out["chunks_local"] > 7000
    for key in ("host", "dev"):
        assert out[key][2].startswith("cuda"), out[key][2]           # the gathered result lives where RCCL put it
        assert out[key][0].shape == t.shape and (out[key][0] == t).all() and np.array_equal(out[key][1], s), key
    assert all(m[2].startswith("cuda") for m in out["many"])
    _check_many(rv, out, t, s)


@pytest.mark.parametrize("enc_depth,dec_depth,attention", [(2, 1, "luong"), (3, 2, "luong"), (2, 1, "bahdanau")])
def test_load_weights_from_tf_checkpoint_prefix(rv, tmp_path, enc_depth, dec_depth, attention):
    """`Basecaller.load_weights(prefix)` as the evaluators call it (ravvent_performance_evaluator.py:107, ravvent.py:57-70) on a live
    handle: a TF-format tensor bundle with the reference model's variable paths (+ optimizer slots and counters, which a
    ModelCheckpoint file also carries) gives byte-identical calls to `set_weights_flat` of the same arrays; a checkpoint that lacks
    a variable, or holds one of the wrong shape, raises and names it."""
    ck = rv.checkpoint
    mk = lambda: rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, encoder_depth=enc_depth, decoder_depth=dec_depth,
                               attention_type=attention, honor_attention_type=True, max_batch=32)
    a, b = mk(), mk()
    flat = rv.weights.init_weights(a.cfg, seed=7, gain=3.0)
    flat["b_fc"][3:7] += 1.5; flat["b_fc"][a.cfg.end_token] -= 1.0
    a.set_weights_flat(flat)
    paths = ck.variable_paths(a.cfg)
    tensors = {paths[n] + "/.ATTRIBUTES/VARIABLE_VALUE": v for n, v in flat.items()}
    tensors["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"] = np.array(77, np.int64)
    tensors["decoder/fc/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE"] = np.ones((128, 7), np.float32)
    tensors["decoder/fc/kernel/.OPTIMIZER_SLOT/optimizer/v/.ATTRIBUTES/VARIABLE_VALUE"] = np.ones((128, 7), np.float32)
    tensors["save_counter/.ATTRIBUTES/VARIABLE_VALUE"] = np.array(3, np.int64)
    prefix = str(tmp_path / "model_chp")
    ck.write_tensor_bundle(prefix, tensors)
    assert b.load_weights(prefix) is b
    raw, ev, _ = rv.synthetic.make_slab(20, 90, 14, seed=enc_depth)
    ta, sa = a.beam_search_prediction((raw, ev), 5, 20)
    tb, sb = b.beam_search_prediction((raw, ev), 5, 20)
    assert ta.shape == tb.shape and (ta.numpy() == tb.numpy()).all() and np.array_equal(sa.numpy(), sb.numpy())
    assert any(len(s) > 3 for s in a.tokens_to_nuc_sequences(ta))
    ga, la = a.greedy_search_prediction((raw, ev), 20)
    gb, lb = b.greedy_search_prediction((raw, ev), 20)
    assert (ga.numpy() == gb.numpy()).all() and np.array_equal(la.numpy(), lb.numpy())
    # error paths on the live handle: a missing variable / a wrong shape are named, and the handle keeps its weights
    few = {k: v for k, v in tensors.items() if "encoder_raw/rnn_layers/1/forward_layer/cell/recurrent_kernel" not in k}
    ck.write_tensor_bundle(str(tmp_path / "few"), few)
    with pytest.raises(KeyError, match="enc_raw.1.fwd.U"):
        b.load_weights(str(tmp_path / "few"))
    bad = dict(tensors); bad[paths["W_mem"] + "/.ATTRIBUTES/VARIABLE_VALUE"] = np.zeros((256, 64), np.float32)
    ck.write_tensor_bundle(str(tmp_path / "bad"), bad)
    with pytest.raises(ValueError, match="shape"):
        b.load_weights(str(tmp_path / "bad"))
    tb2, _ = b.beam_search_prediction((raw, ev), 5, 20)
    assert (tb2.numpy() == tb.numpy()).all()
    a.close(); b.close()


def _saturating_encoder(rv, cfg, flat, layers=(0, 1)):
    """Half of every encoder cell's units are driven to |h| = 1 - O(1e-10) by their biases: input and output gates wide open
    (bias 20), forget gate at 0.92 (bias 2.5), candidate at +-1 (bias +-10): c settles at +-13 and h = sigmoid(20) tanh(c).
    The other half stay Keras-default, so the layers still depend on their input and the problem stays well-conditioned
    (an LSTM that INTEGRATES its input into saturation is ill-conditioned in any fp32 arithmetic: the numpy fp32 twin of the
    oracle is then 1e-3 off the fp64 oracle -- tried and dropped)."""
    sgn = np.where(np.arange(64) % 2 == 0, 10.0, -10.0).astype(np.float32)
    for e in ("raw", "event"):
        for d in ("fwd", "bwd"):
            for l in layers:
                b = flat[f"enc_{e}.{l}.{d}.b"]
                b[0:64] = 20.0; b[128:192] = 2.5; b[256:320] = sgn; b[384:448] = 20.0


def test_split_operands_saturated_activations(rv, oracle):
    """Split-f16 operands where the activation side is at its bound: in both encoder layers half of the units saturate (|h| within
    2^-20 of 1: the 2^14-scaled value rounds UP to 2^14 in its high f16 part and the low part is negative), so the layer-1
    input projection and the memory projection multiply A operands at the top of their range, next to ordinary ones.
    enc_output within 1e-4 of the fp64 oracle and no further from it than the exact-f32 MFMA path of the same library; the memory
    projection of the GPU's OWN enc_output (no recurrence in between) against the fp64 product, with a key column x 100 and one
    x 1e-5, to 1e-6 of its range."""
    B, Tr, Te = 40, 150, 20
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=Tr, max_event_len=Te)
    flat = rv.weights.init_weights(bc.cfg, seed=13)
    _saturating_encoder(rv, bc.cfg, flat)
    flat["W_mem"][:, 5] *= 100.0; flat["W_mem"][:, 9] *= 1e-5
    bc.set_weights_flat(flat)
    w = rv.weights.flat_to_nested(bc.cfg, flat)
    raw, ev, _ = rv.synthetic.make_slab(B, Tr, Te, seed=3)
    e64, _ = oracle.encode_input(w, raw, ev, "joint", 0.0, np.float64)
    e32, _ = oracle.encode_input(w, raw, ev, "joint", 0.0, np.float32)
    assert (np.abs(e64) > 1.0 - 2.0 ** -20).mean() > 0.3                  # the regime was reached
    wmp = np.concatenate([flat["W_mem"], flat["W_att"][128:384]], axis=1).astype(np.float64)
    err, merr = {}, {}
    for split in (2, 0, 3):
        # 2 / 0: the packed-FMA recurrence with its fused projection on split-f16 / f32 MFMAs, the memory projection likewise;
        # 3: the default path -- matrix-pipe recurrence (h itself a split-f16 operand at its bound) + split-f16 projection GEMM
        bc.set_option("wide_recurrence", 1 if split == 3 else 0)
        bc.set_option("split_projection", 2 if split == 3 else split)
        bc.beam_search_prediction((raw, ev), 3, 4)
        enc = bc.get_tensor("enc_output").reshape(B, Tr + Te, 256)
        assert np.isfinite(enc).all() and np.abs(enc).max() <= 1.0
        err[split] = float(np.abs(enc - e64).max())
        ref = enc.astype(np.float64) @ wmp
        mem = bc.get_tensor("projected_memory").reshape(B, Tr + Te, 256)
        merr[split] = float(np.abs(mem - ref).max() / np.abs(ref).max())
    twin = float(np.abs(e32 - e64).max())
    print(f"saturated encoder: |enc_output - fp64| matrix-pipe recurrence {err[3]:.2e}, FMA recurrence + split-f16 projection {err[2]:.2e}, "
          f"+ f32 MFMA projection {err[0]:.2e}, numpy fp32 twin {twin:.2e}; memory projection / range: split {merr[2]:.2e}, f32 {merr[0]:.2e}")
    assert err[2] < TOL and err[2] <= 2.0 * err[0] + 1e-6, (err, twin)
    assert err[3] < TOL and err[3] <= 2.0 * err[0] + 1e-6, (err, twin)
    assert merr[2] <= 1e-6 and merr[3] <= 1e-6 and merr[2] <= 1.5 * merr[0] + 2e-7, merr
    bc.close()


def test_split_operands_adversarial_attention(rv, oracle):
    """The default decode (matrix_attention = 1: Luong scores and context as three exact f16 part products each) against the fp64
    oracle where the split is under the most stress: W_mem and W_att eight times the Keras scale (scores of tens: near one-hot
    alignments), one key column x 100 and one x 1e-5 (a 1e7 range inside one f16-scaled matrix, keys up to ~2,400), one column of
    the attention layer's context rows x 100, a chunk with a SINGLE unmasked step (the alignment is exactly one-hot) and one with
    three.  The decoder cell is left at Keras scale on purpose (its own split operands are stressed in
    test_split_operands_adversarial_cell): at x 8 the decode turns chaotic -- the numpy fp32 twin of the oracle then leaves the fp64
    oracle by > 1e-4 at the third step, so a bound there would test nothing.  Per-step logits of every chunk whose beam order equals the fp64 one: within 1e-4; any other chunk must
    pass through a near-tie in exact arithmetic."""
    from test_parity_gpu import _near_tie_gap
    B, Tr, Te, W, L = 12, 120, 20, 5, 16
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=Tr, max_event_len=Te)
    flat = rv.weights.init_weights(bc.cfg, seed=13)
    flat["W_mem"] *= 128.0; flat["W_att"] *= 8.0; flat["W_fc"] *= 0.125
    flat["W_mem"][:, 5] *= 100.0
    flat["W_mem"][:, 9] *= 1e-5
    flat["W_att"][128:, 17] *= 12.5
    flat["b_fc"][bc.cfg.end_token] = -2.0
    bc.set_weights_flat(flat)
    w = rv.weights.flat_to_nested(bc.cfg, flat)
    raw, ev, _ = rv.synthetic.make_slab(B, Tr, Te, seed=3)
    raw[0, 1:] = 0.0; ev[0] = 0.0                            # chunk 0: one unmasked step
    raw[1, 2:] = 0.0; ev[1, 1:] = 0.0                        # chunk 1: two raw steps + one event
    bc.set_option("persist_taps", 1)
    bc.set_option("profile", 1)
    out = {}
    for mx in (1, 0):
        bc.set_option("matrix_attention", mx)
        tok, sc = bc.beam_search_prediction((raw, ev), W, L)
        assert "dec_persist" in bc.profile()                # the one-launch decode ran
        S = tok.shape[1]
        out[mx] = (tok.numpy().copy(), sc.numpy().copy(), bc.get_tensor("step_logits").reshape(S, B, W, 7),
                   bc.get_tensor("parent_ids").reshape(S, B, W), bc.get_tensor("chunk_steps").astype(int))
    taps = {}
    ot, osc = oracle.beam_search(w, bc.cfg.oracle_cfg(), raw, ev, W, L, dtype=np.float64, taps=taps)
    assert np.abs(taps["keys"]).max() > 1000.0 and taps["step_alignments"][:, 0].max() == 1.0 and taps["step_alignments"].max(-1).mean() > 0.6
    assert np.abs(bc.get_tensor("enc_output").reshape(B, Tr + Te, 256) - taps["enc_output"]).max() < TOL
    end = bc.cfg.oracle_cfg()["end_token"]
    for mx in (1, 0):
        tok, sc, lg, par, cs = out[mx]
        n_tie, worst = 0, 0.0
        for b in range(B):
            n = min(cs[b], ot.shape[1])
            if (par[:n, b] == taps["parent_ids"][:n, b]).all() and (tok[b, :ot.shape[1]] == ot[b]).all():
                e = float(np.abs(lg[:n, b] - taps["step_logits"][:n, b]).max())
                worst = max(worst, e)
                assert e < TOL, (mx, b, e)
                assert np.abs(sc[b, :ot.shape[1]] - osc[b]).max() < TOL, (mx, b)
            else:
                gap = _near_tie_gap(oracle, taps["step_logits"][:, b], W, end)
                assert gap < TOL, f"matrix_attention={mx}, chunk {b}: differs from fp64 with no near-tie (smallest gap {gap:.3e})"
                n_tie += 1
        print(f"matrix_attention={mx}: max |logits - fp64| {worst:.2e} over {B - n_tie} chunks, {n_tie} near-tie chunk(s)")
        assert n_tie <= B // 4, n_tie
    bc.close()


def test_split_operands_adversarial_cell(rv, oracle):
    """The decoder cell's product on the matrix pipe (matrix_cell = 1, the default: [ctx' | h] . [W_a ; U + A_h W_a] as three exact f16
    part products, ONE power-of-two scale for the whole kernel, the ctx' rows divided by the scale of the context image) where that
    split is under stress: the attention layer's context rows x 32 (|ctx'| of several units against |h| < 1, and a context-image bound
    of hundreds: the two halves of the kernel end up 2^5 apart in the image), one gate column of the cell's attention rows x 50 and one x 1e-4, one recurrent row x 30, one x 1e-5 (a
    1e6 range inside the one scaled tensor).  Four steps only -- with these weights the cell is a chaotic map and any fp32 evaluation
    leaves fp64 soon after -- per-step logits of every chunk whose beam order equals the fp64 one within 1e-4, both cell forms; any
    other chunk must pass through a near-tie in exact arithmetic."""
    from test_parity_gpu import _near_tie_gap
    B, Tr, Te, W, L = 10, 150, 25, 5, 5
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=Tr, max_event_len=Te)
    flat = rv.weights.init_weights(bc.cfg, seed=31)
    flat["W_att"][128:] *= 32.0
    flat["dec_cells.0.W"][7:, 5] *= 50.0
    flat["dec_cells.0.W"][7:, 300] *= 1e-4
    flat["dec_cells.0.U"][3, :] *= 30.0
    flat["dec_cells.0.U"][77, :] *= 1e-5
    flat["W_fc"] *= 0.25
    flat["b_fc"][bc.cfg.end_token] = -2.0
    bc.set_weights_flat(flat)
    w = rv.weights.flat_to_nested(bc.cfg, flat)
    raw, ev, _ = rv.synthetic.make_slab(B, Tr, Te, seed=8)
    bc.set_option("persist_taps", 1)
    bc.set_option("profile", 1)
    out = {}
    for mc in (1, 0):
        bc.set_option("matrix_cell", mc)
        tok, sc = bc.beam_search_prediction((raw, ev), W, L)
        assert "dec_persist" in bc.profile()
        S = tok.shape[1]
        out[mc] = (tok.numpy().copy(), sc.numpy().copy(), bc.get_tensor("step_logits").reshape(S, B, W, 7),
                   bc.get_tensor("parent_ids").reshape(S, B, W), bc.get_tensor("chunk_steps").astype(int))
    taps = {}
    ot, osc = oracle.beam_search(w, bc.cfg.oracle_cfg(), raw, ev, W, L, dtype=np.float64, taps=taps)
    assert np.abs(taps["enc_output"] @ flat["W_att"][128:].astype(np.float64)).max() > 4.0    # the regime: context images of several units
    end = bc.cfg.oracle_cfg()["end_token"]
    for mc in (1, 0):
        tok, sc, lg, par, cs = out[mc]
        n_tie, worst = 0, 0.0
        for b in range(B):
            n = min(cs[b], ot.shape[1])
            if (par[:n, b] == taps["parent_ids"][:n, b]).all() and (tok[b, :ot.shape[1]] == ot[b]).all():
                e = float(np.abs(lg[:n, b] - taps["step_logits"][:n, b]).max())
                worst = max(worst, e)
                assert e < TOL, (mc, b, e)
                assert np.abs(sc[b, :ot.shape[1]] - osc[b]).max() < TOL, (mc, b)
            else:
                gap = _near_tie_gap(oracle, taps["step_logits"][:, b], W, end)
                assert gap < TOL, f"matrix_cell={mc}, chunk {b}: differs from fp64 with no near-tie (smallest gap {gap:.3e})"
                n_tie += 1
        print(f"matrix_cell={mc}: max |logits - fp64| {worst:.2e} over {B - n_tie} chunks, {n_tie} near-tie chunk(s)")
        assert n_tie <= B // 4, n_tie
    bc.close()
