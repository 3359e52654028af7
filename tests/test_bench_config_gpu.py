"""Parity of what bench.py actually runs, and the sweeps that used to live in tools/ ("next" 1):

* the bench's product configuration -- C3-size slabs (256 x 300 + 30, beam 5, L 48) streamed through the asynchronous calls at depth
  10, library-allocated outputs / caller-provided output tensors / raw output addresses inside one gather buffer
  (dist.sharded_beam_search_many's form) -- byte-identical to the synchronous call per slab, twenty DISTINCT slabs, and one slab
  against the C port with every differing row settled by the fp64 oracle;
* a synchronous call while the handle's own context holds an uncollected ticket;
* the random sweep of tools/stress_paths.py (six forms of the path against each other, asynchronous calls, greedy) with every token
  OR score difference routed through the fp64 re-decode of test_parity_gpu._explain_mismatches -- incl. the case of
  gpurun_out/r3b_stress.log (seed 123, case 88) that differed by 9.2e-4 in a score and was printed, not counted;
* adversarial recurrent kernels for the matrix-pipe recurrence's `Ua` image (per-gate-column power-of-two factors, lstm_mx.hip).

Reference calls these stand behind: /root/reference/ravvent_performance_evaluator.py:51-55, /root/reference/basecaller.py:19-32."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _same(got, want):
    t, s = got
    return tuple(t.shape) == want[0].shape and (np.asarray(t.cpu()) == want[0]).all() and np.array_equal(np.asarray(s.cpu()), want[1])


def test_bench_configuration_async_depth10_matches_synchronous(rv, oracle):
    """bench.py's timed region: C3 slabs through rv_beam_search_submit_dev / collect_dev, ten in flight on the handle's contexts."""
    import torch
    from oracle import cpu_port
    from test_parity_gpu import _emitting_flat, _explain_mismatches, _assert_calls_are_strings
    B, T_r, T_e, W, L, K, depth = 256, 300, 30, 5, 48, 20, 10
    steps = L - 1
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e,
                       max_output_len=L)
    flat = _emitting_flat(rv, bc.cfg)
    bc.set_weights_flat(flat)
    host = [rv.synthetic.make_slab(B, T_r, T_e, seed=1000 + k)[:2] for k in range(K)]
    dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(e).cuda()) for r, e in host]
    ref = []
    for x in dev:                                             # the synchronous call, fresh outputs per slab
        t, s = bc.beam_search_prediction(x, W, L)
        ref.append((t.cpu().numpy().copy(), s.cpu().numpy().copy()))
    assert len({r[0].tobytes() for r in ref}) == K          # the slabs really differ
    _assert_calls_are_strings(rv, bc, ref[0][0], "C3 stream")
    bc.set_async_depth(depth)

    # (1) library-allocated outputs, the form bench.py streams
    outs = list(bc.beam_search_stream(dev, W, L))
    assert len(outs) == K and all(_same(o, r) for o, r in zip(outs, ref))
    # (2) the same with bench.py's reuse_output_buffers flag set and the caller's own output tensors, one pair per slab
    bc.reuse_output_buffers = True
    mine = [(torch.full((B, steps), -7, dtype=torch.int32, device="cuda"), torch.full((B, steps), np.nan, dtype=torch.float32, device="cuda"))
            for _ in range(K)]
    outs = list(bc.beam_search_stream(dev, W, L, outs=mine))
    assert all(_same(o, r) for o, r in zip(outs, ref))
    for (t, s), r in zip(mine, ref):                          # ... and they landed in the caller's tensors
        S = r[0].shape[1]
        assert (t[:, :S].cpu().numpy() == r[0]).all() and np.array_equal(s[:, :S].cpu().numpy(), r[1])
    # (3) raw addresses inside ONE buffer, as dist.sharded_beam_search_many submits: [K][token plane | score-bit plane]
    packed = torch.full((K, 2, B, steps), -1, dtype=torch.int32, device="cuda")
    base, plane = packed.data_ptr(), 4 * B * steps
    queue, S_of = [], []
    for k, x in enumerate(dev):
        if len(queue) >= depth:
            S_of.append(bc.collect(queue.pop(0)))
        queue.append(bc.submit_beam_search(x, W, L, out_ptrs=(base + 2 * k * plane, base + (2 * k + 1) * plane)))
    while queue:
        S_of.append(bc.collect(queue.pop(0)))
    got = packed.cpu().numpy()
    for k, r in enumerate(ref):
        S = r[0].shape[1]
        assert S_of[k] == S and (got[k, 0, :, :S] == r[0]).all() and np.array_equal(got[k, 1, :, :S].view(np.float32), r[1]), k
    # (4) bench.py feeds ONE slab to every step: ten copies of the same slab in flight at once
    outs = list(bc.beam_search_stream([dev[3]] * K, W, L))
    assert all(_same(o, ref[3]) for o in outs)

    # a synchronous call while every context holds a ticket is refused and orphans nothing; with the handle's OWN context busy
    # and another one idle it runs there (it used to fail on context 0 and drop that context's ticket)
    tickets = [bc.submit_beam_search(dev[i], W, L) for i in range(depth)]
    with pytest.raises(rv._capi.RavventHipError, match="every slab context"):
        bc.beam_search_prediction(dev[11], W, L)
    assert _same(bc.collect(tickets[4]), ref[4])              # frees a child context; ticket 0 still sits on the handle's own
    assert _same(bc.beam_search_prediction(dev[11], W, L), ref[11])
    t_h, s_h = bc.beam_search_prediction(host[12], W, L)      # host-buffer entry point as well
    assert _same((t_h, s_h), ref[12])
    for i in (0, 1, 2, 3, 5, 6, 7, 8, 9):
        assert _same(bc.collect(tickets[i]), ref[i]), i

    # slab 0 against the fp32 C port; rows that differ are re-decoded by the fp64 oracle (near-tie or fail)
    raw0, ev0 = host[0]
    ctok, csc = cpu_port.run(bc.cfg.oracle_cfg(), 2, 7, rv.weights.pack(bc.cfg, flat), raw0, ev0, W, L)
    assert ctok.shape == ref[0][0].shape
    _, n_bad = _explain_mismatches(rv, oracle, bc, flat, "joint", raw0, ev0, W, L, ref[0][0], ctok, "C3 stream slab 0", ref[0][1], csc)
    print(f"C3 stream: {K} distinct slabs x 4 output forms byte-identical to the synchronous call; slab 0 vs C port: {n_bad} rows explained by fp64")
    bc.close()


# ------------------------------------------------------------------------------------------------------------------------------
def _explain_greedy(oracle, w, cfg, mode, raw, ev, L, a, b, tag):
    """Greedy rows on which two forms of the path disagree: the fp64 oracle decodes the row alone; at the first step where the two
    forms' tokens differ the fp64 logits' two largest entries must be closer than 1e-4 (either argmax is then legitimate, and the
    rows diverge from there), else fail.  Rows whose tokens agree must agree in their logits to 1e-4."""
    ta, la, tb, lb = a[0], a[1], b[0], b[1]
    S = min(ta.shape[1], tb.shape[1])
    rows = (ta[:, :S] == tb[:, :S]).all(axis=1)
    if S and rows.any():
        assert np.abs(la[rows, :S] - lb[rows, :S]).max() < TOL, tag
    bad = np.nonzero(~rows)[0]
    assert ta.shape[1] == tb.shape[1] or bad.size, f"{tag}: greedy step counts {ta.shape[1]} / {tb.shape[1]} differ with every row equal"
    assert bad.size <= max(1, ta.shape[0] // 20), f"{tag}: {bad.size} of {ta.shape[0]} greedy rows differ"
    for r in bad:
        t = int(np.argmax(ta[r, :S] != tb[r, :S]))
        _, olg = oracle.greedy_search(w, cfg, raw[r:r + 1] if mode != "event" else None, ev[r:r + 1] if mode != "raw" else None, L,
                                      dtype=np.float64)
        if t >= olg.shape[1]:      # the row alone stops before the step in question: everything up to there agreed, which is the claim
            continue
        top = np.sort(olg[0, t])[::-1]
        assert top[0] - top[1] < TOL, f"{tag}: greedy row {r} flips at step {t} with fp64 margin {top[0] - top[1]:.3e}"
    return int(bad.size)


FORMS = {"pf": (1, 1, 1, 1), "sf": (0, 1, 1, 1), "pu": (1, 0, 0, 1), "fm": (1, 1, 0, 1), "fc": (1, 1, 1, 0), "p8": (1, 1, 2, 1)}


def _sweep_case(rv, oracle, mode, enc_d, dec_d, B, T_r, T_e, W, L, att, wseed, end_bias, xseed, tag):
    """One case of the sweep: the default path (pf), the per-step decode kernels (sf), the packed-FMA recurrences with an unfused (pu) /
    fused (fm) projection, the decoder cell's product on packed FMAs (fc) and the eight-chunk matrix-pipe recurrences (p8) on the same slab; every pair (pf, other) must agree row by
    row in tokens AND scores, or the fp64 oracle must explain the row for BOTH forms (test_parity_gpu._explain_mismatches)."""
    from test_parity_gpu import _explain_mismatches
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, mode, 0.0, encoder_depth=enc_d, decoder_depth=dec_d, max_batch=B,
                       attention_type=att, honor_attention_type=True)
    flat = rv.weights.init_weights(bc.cfg, seed=wseed)
    flat["b_fc"][bc.cfg.end_token] = end_bias
    bc.set_weights_flat(flat)
    w = rv.weights.flat_to_nested(bc.cfg, flat)
    raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=xseed, max_raw_pad=min(15, T_r - 1), max_event_pad=min(10, T_e - 1))
    x = (raw, ev) if mode == "joint" else (raw if mode == "raw" else ev)
    out = {}
    for key, (persist, fuse, wide, mcell) in FORMS.items():
        bc.set_option("persistent_decode", persist); bc.set_option("fused_projection", fuse); bc.set_option("wide_recurrence", wide)
        bc.set_option("matrix_cell", mcell)
        t, s = bc.beam_search_prediction(x, W, L)
        if key == "pf":      # asynchronous calls: the same slab three times in flight, each byte-identical to the synchronous result
            bc.set_async_depth(3)
            for ta, sa in bc.beam_search_stream([x, x, x], W, L):
                assert ta.shape == t.shape and (ta.numpy() == t.numpy()).all() and np.array_equal(sa.numpy(), s.numpy()), f"{tag}: asynchronous != synchronous"
        g, lg = bc.greedy_search_prediction(x, L)
        out[key] = (t.numpy().copy(), s.numpy().copy(), g.numpy().copy(), lg.numpy().copy())
    n_rows = n_greedy = 0
    for other in ("sf", "pu", "fm", "fc", "p8"):
        a, b = out["pf"], out[other]
        assert a[0].shape == b[0].shape, f"{tag} vs {other}: beam shapes {a[0].shape} / {b[0].shape}"
        if a[0].size:
            # both directions: the first form named is the one whose rows the fp64 oracle must confirm (or find a near-tie for)
            _, n1 = _explain_mismatches(rv, oracle, bc, flat, mode, raw, ev, W, L, a[0], b[0], f"{tag} pf vs {other}", a[1], b[1])
            _explain_mismatches(rv, oracle, bc, flat, mode, raw, ev, W, L, b[0], a[0], f"{tag} {other} vs pf", b[1], a[1])
            n_rows += n1
        if a[2].size or b[2].size:
            n_greedy += _explain_greedy(oracle, w, bc.cfg.oracle_cfg(), mode, raw, ev, L, (a[2], a[3]), (b[2], b[3]), f"{tag} greedy pf vs {other}")
    bc.close()
    return n_rows, n_greedy


def test_stress_case_88_score_difference_is_explained(rv, oracle):
    """gpurun_out/r3b_stress.log (round 3, RV_STRESS_SEED=123): `score diff: max 9.232e-04 ... vs fc: rows equal 1.0000 scores ok
    False` -- the matrix-pipe cell product against the packed-FMA cell product, tokens equal, top-1 scores 9x the 1e-4 bar apart, and
    tools/stress_paths.py printed it without counting it.  Same case here (the 89th draw of that generator), every differing row
    through the fp64 oracle: the row's decode passes through a near-tie at the beam cut (a different fifth beam changes the later,
    un-back-traced top-1 scores: SURVEY.md A.5), or this test is red."""
    n_rows, n_greedy = _sweep_case(rv, oracle, "joint", 2, 1, 169, 18, 29, 4, 29, "luong", 425, -0.3332263726534489, 88, "stress seed 123 case 88")
    # (On the round-3 kernels, and on round 4's until the encoder's cell update changed its rounding, one row differed here -- tokens equal,
    #  scores 9.2e-4 apart -- and the fp64 re-decode showed the near-tie; a later rounding change can make the two forms agree outright.)
    print(f"case 88: {n_rows} beam rows and {n_greedy} greedy rows differed between forms, all explained by the fp64 oracle")


@pytest.mark.parametrize("seed", [123, 7])
def test_stress_sweep_every_difference_explained(rv, oracle, seed):
    """tools/stress_paths.py as a test: random modes / depths / shapes / beams / attention types, 24 cases per seed (seed 123 = the
    first 24 draws of the round-3 sweep)."""
    rng = np.random.default_rng(seed)
    tot = [0, 0]
    for case in range(24):
        mode = ("joint", "raw", "event")[int(rng.integers(0, 3))]
        enc_d, dec_d = int(rng.integers(1, 4)), int(rng.integers(1, 3))
        B, T_r, T_e = int(rng.integers(1, 400)), int(rng.integers(1, 301)), int(rng.integers(1, 46))
        W = int(rng.integers(1, 9 if dec_d == 1 else 6)); L = int(rng.integers(2, 40))
        att = ("luong", "bahdanau")[int(rng.integers(0, 2))]
        wseed = int(rng.integers(0, 1000)); end_bias = float(rng.uniform(-1, 2))
        r, g = _sweep_case(rv, oracle, mode, enc_d, dec_d, B, T_r, T_e, W, L, att, wseed, end_bias, case,
                           f"seed {seed} case {case} {(mode, att, enc_d, dec_d, B, T_r, T_e, W, L)}")
        tot[0] += r; tot[1] += g
    print(f"seed {seed}: 24 cases x 6 forms; {tot[0]} beam rows and {tot[1]} greedy rows differed between forms, all explained by the fp64 oracle")


# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Tr,Te,col_gain", [(60, 20, 50.0), (150, 25, 8.0)])
def test_matrix_pipe_recurrence_adversarial_recurrent_kernel(rv, oracle, Tr, Te, col_gain):
    """The encoder's recurrent kernel as `k_lstm_rec_mx` holds it: U^T as split-f16 A fragments with one power-of-two factor per
    gate COLUMN of U (row of U^T; lstm_mx.hip, rv_load_weights).  Stress for that image in every recurrent kernel of both encoders and
    both layers: one recurrent row x 30 (every column's largest element then sits in that row, the others lose 5 bits of the
    high part's range), one row x 1e-5 (deep in the low part / f16 subnormals of its columns), and two gate columns x 50 (own
    factors 2^6 apart from their neighbours'; pre-activations of tens).  enc_output of the matrix-pipe recurrence within 1e-4 of
    the fp64 oracle and no further from it than twice the packed-FMA kernels (exact f32 products) of the same library.
    A gate whose recurrent column carries a gain of 50 amplifies ANY fp32 rounding step after step: at 150 + 25 steps the numpy fp32
    twin of the oracle is 1.8e-5 from fp64 and the packed-FMA kernels -- exact f32 products, hardware exp / rcp -- 1.4e-4 (the matrix
    pipe: 7.7e-5; first run of this test), so the x 50 case runs on 60 + 20 steps and the long case with a gain of 8."""
    B = 37
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=Tr, max_event_len=Te)
    flat = rv.weights.init_weights(bc.cfg, seed=13)
    for e in ("raw", "event"):
        for l in (0, 1):
            for d in ("fwd", "bwd"):
                U = flat[f"enc_{e}.{l}.{d}.U"]
                U[17, :] *= 30.0; U[90, :] *= 1e-5; U[:, 128 + 40] *= col_gain; U[:, 300] *= col_gain
    bc.set_weights_flat(flat)
    w = rv.weights.flat_to_nested(bc.cfg, flat)
    raw, ev, _ = rv.synthetic.make_slab(B, Tr, Te, seed=3)
    e64, _ = oracle.encode_input(w, raw, ev, "joint", 0.0, np.float64)
    e32, _ = oracle.encode_input(w, raw, ev, "joint", 0.0, np.float32)
    twin = float(np.abs(e32 - e64).max())
    err, toks = {}, {}
    bc.set_option("profile", 1)
    for wide in (1, 2, 0):                                            # matrix pipe with 16 / with 8 chunks per workgroup, packed FMA
        bc.set_option("wide_recurrence", wide)
        bc.reset_profile()
        tok, _ = bc.beam_search_prediction((raw, ev), 3, 6)
        assert ("gemm_inproj_raw" in bc.profile()) == bool(wide)      # the matrix-pipe form really ran / did not run
        enc = bc.get_tensor("enc_output").reshape(B, Tr + Te, 256)
        assert np.isfinite(enc).all()
        err[wide] = float(np.abs(enc - e64).max())
        toks[wide] = tok.numpy().copy()
    print(f"adversarial recurrent kernels: max |enc_output - fp64| matrix pipe {err[1]:.2e} (8 chunks per workgroup {err[2]:.2e}), packed FMA {err[0]:.2e}, numpy fp32 twin {twin:.2e}")
    assert err[1] < TOL and err[2] < TOL and err[0] < TOL, (err, twin)
    assert err[1] <= 2.0 * err[0] + 2e-6 and err[2] <= 2.0 * err[0] + 2e-6, (err, twin)
    bc.close()


def test_slab_graph_replay_matches_launches(rv):
    """Option slab_graph: a call on the default path replays as ONE hipGraphLaunch per slab context and call shape, with every kernel
    argument frozen at capture and the caller's input / output addresses read through a table in mapped pinned memory.  Byte-identical
    to the launch-by-launch form for device inputs at EVER-CHANGING addresses (a stale pointer would decode the wrong slab or write the
    wrong tensor), host inputs, the fused post-processing, greedy search, several shapes through one handle, asynchronous and
    synchronous calls; new weights and a changed option take effect (the frozen arguments hold weight-derived scalars)."""
    import torch
    from test_parity_gpu import _emitting_flat
    B, T_r, T_e, W, L = 48, 120, 20, 5, 24
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L)
    flat = _emitting_flat(rv, bc.cfg, seed=5)
    bc.set_weights_flat(flat)
    host = [rv.synthetic.make_slab(n, T_r, T_e, seed=300 + i)[:2] for i, n in enumerate((48, 48, 17, 48, 48, 17, 48, 1))]
    ref = [tuple(a.numpy().copy() for a in bc.beam_search_prediction(x, W, L)) for x in host]
    ref_calls = [bc.beam_search_call_arrays(x, W, L) for x in host]
    ref_greedy = [tuple(a.numpy().copy() for a in bc.greedy_search_prediction(x, L)) for x in host[:3]]
    bc.set_option("slab_graph", 1)
    bc.set_async_depth(3)
    for rep in range(3):       # rep 0 captures (per context and shape), later reps replay; fresh device tensors = fresh addresses every time
        dev = [(torch.from_numpy(r.copy()).cuda(), torch.from_numpy(e.copy()).cuda()) for r, e in host]
        pad = torch.empty(1000 * (rep + 1), device="cuda")        # (shifts the allocator's addresses between reps)
        outs = list(bc.beam_search_stream(dev, W, L))
        assert all(_same(o, r) for o, r in zip(outs, ref)), rep
        outs = list(bc.beam_search_stream(host, W, L))
        assert all(_same(o, r) for o, r in zip(outs, ref)), rep
        for got, want in zip(bc.beam_search_stream(host, W, L, calls=True), ref_calls):
            assert all(np.array_equal(g, w_) for g, w_ in zip(got, want)), rep
        for x, r in zip(dev, ref):                               # synchronous calls replay too
            assert _same(bc.beam_search_prediction(x, W, L), r), rep
        for x, r in zip(host[:3], ref_greedy):
            g, lg = bc.greedy_search_prediction(x, L)
            assert (g.numpy() == r[0]).all() and np.array_equal(lg.numpy(), r[1]), rep
        del pad
    # an option that changes the kernels, then new weights: the captured graphs are rebuilt, not replayed stale
    bc.set_option("matrix_cell", 0)
    t0, s0 = bc.beam_search_prediction(host[0], W, L)
    bc.set_option("slab_graph", 0)
    t1, s1 = bc.beam_search_prediction(host[0], W, L)
    assert (t0.numpy() == t1.numpy()).all() and np.array_equal(s0.numpy(), s1.numpy())
    bc.set_option("matrix_cell", 1); bc.set_option("slab_graph", 1)
    flat2 = _emitting_flat(rv, bc.cfg, seed=6)
    bc.set_weights_flat(flat2)
    t2, s2 = bc.beam_search_prediction(host[0], W, L)
    bc.set_option("slab_graph", 0)
    t3, s3 = bc.beam_search_prediction(host[0], W, L)
    assert (t2.numpy() == t3.numpy()).all() and np.array_equal(s2.numpy(), s3.numpy()) and not np.array_equal(s2.numpy(), ref[0][1])
    bc.close()
