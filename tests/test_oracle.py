"""CPU tests of the oracle (the checker itself): numpy fp64 restatement vs torch.nn.LSTM, vs
hand-computed micro cases for the beam step / gather_tree, vs its C twin, and vs the committed
golden fixtures.  PARITY UNPINNED against TensorFlow (absent here) -- see oracle/ravvent_oracle.py."""
import importlib.util
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden_mod():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_lstm_stack_matches_torch(rv, oracle):
    """Every BiLSTM layer incl. the state chaining of Encoder.call (basecaller.py:51-57) against
    torch.nn.LSTM (same i,f,g,o gate order; b_hh = 0)."""
    cfg = rv.RvConfig(enc_units=24, dec_units=24, enc_depth=3)
    w = rv.weights.flat_to_nested(cfg, rv.weights.init_weights(cfg, seed=3))
    raw, ev, _ = rv.synthetic.make_slab(4, 17, 7, seed=1)
    for enc, x in (("enc_raw", raw), ("enc_event", ev)):
        out, st = oracle.encoder(np.asarray(x, np.float64), oracle._cast_layers(w[enc], np.float64))
        xt, states = torch.tensor(x, dtype=torch.float64), None
        for lw in w[enc]:
            m = torch.nn.LSTM(lw["fwd"][0].shape[0], 24, batch_first=True, bidirectional=True).double()
            with torch.no_grad():
                for sfx, d in (("", "fwd"), ("_reverse", "bwd")):
                    W, U, b = lw[d]
                    getattr(m, "weight_ih_l0" + sfx).copy_(torch.tensor(W.T.astype(np.float64)))
                    getattr(m, "weight_hh_l0" + sfx).copy_(torch.tensor(U.T.astype(np.float64)))
                    getattr(m, "bias_ih_l0" + sfx).copy_(torch.tensor(b.astype(np.float64)))
                    getattr(m, "bias_hh_l0" + sfx).zero_()
                xt, states = m(xt) if states is None else m(xt, states)
        assert np.abs(out - xt.numpy()).max() < 1e-12
        assert np.abs(st[0] - states[0][0].numpy()).max() < 1e-12      # h_f
        assert np.abs(st[3] - states[1][1].numpy()).max() < 1e-12      # c_b


def test_beam_step_micro_case(oracle):
    """W=2, V=3, end=1: hand-computed expansion incl. -inf start beam, tie order, finished rows."""
    logits = np.log(np.array([[[0.5, 0.3, 0.2], [0.2, 0.2, 0.6]]]))
    lp0 = np.array([[0.0, -np.inf]])
    top, word, parent, lp, fin, ln = oracle.beam_search_step(logits, lp0, np.zeros((1, 2), bool), np.zeros((1, 2), np.int64), 1)
    assert word.tolist() == [[0, 1]] and parent.tolist() == [[0, 0]]
    assert np.allclose(top, np.log([[0.5, 0.3]]))
    assert fin.tolist() == [[False, True]] and ln.tolist() == [[1, 1]]
    # next step: beam 1 is finished -> only its end-token candidate survives, score unchanged
    top2, word2, parent2, _, fin2, ln2 = oracle.beam_search_step(logits, lp, fin, ln, 1)
    # candidates: beam0: log.5+log(.5,.3,.2) ; beam1(finished): [min, log.3+0, min]
    exp = sorted([(np.log(0.5) + np.log(0.5), 0, 0), (np.log(0.3), 1, 1), (np.log(0.5) + np.log(0.3), 1, 0)], reverse=True)[:2]
    assert np.allclose(top2[0], [e[0] for e in exp])
    assert word2[0].tolist() == [e[1] for e in exp] and parent2[0].tolist() == [e[2] for e in exp]
    assert ln2[0].tolist() == [2 if e[2] == 0 else 1 for e in exp]
    # exact ties -> lower flat index first (tf.math.top_k)
    tie = np.zeros((1, 2, 3))
    _, w3, p3, *_ = oracle.beam_search_step(tie, np.zeros((1, 2)), np.zeros((1, 2), bool), np.zeros((1, 2), np.int64), 1)
    assert w3.tolist() == [[0, 1]] and p3.tolist() == [[0, 0]]


def test_gather_tree_micro_case(oracle):
    ids = np.array([[[3, 4]], [[5, 1]], [[6, 3]]], np.int32)        # [S=3, B=1, W=2]
    par = np.array([[[0, 0]], [[1, 0]], [[0, 1]]], np.int32)
    out = oracle.gather_tree(ids, par, np.array([3]), end_token=1)
    # beam0: t2 id 6 parent 0 -> t1 id 5 parent 1 -> t0 id 4 ; beam1: t2 id 3 parent 1 -> t1 id 1 (end) ...
    assert out[:, 0, 0].tolist() == [4, 5, 6]
    assert out[:, 0, 1].tolist() == [3, 1, 1]          # everything after the first end token is end
    out2 = oracle.gather_tree(ids, par, np.array([2]), end_token=1)   # max length 2: step 2 stays end
    assert out2[:, 0, 0].tolist() == [4, 5, 1]


def test_strings_and_probs(rv, oracle):
    tok = np.array([[3, 4, 5, 6, 1, 1], [2, 6, 0, 3, 1, 4]])
    assert oracle.tokens_to_nuc_sequences(tok) == ["ACGT", "TAC"]
    assert rv.data_loader.tokens_to_strings(tok) == ["ACGT", "TAC"]
    sc = np.log(np.array([[0.5, 0.25, 0.125]]))
    assert np.allclose(oracle.calc_prob_logits_beam_search_scores(sc), [[0.5, 0.5, 0.5]])
    assert np.allclose(rv.utils.calc_prob_logits_beam_search_scores(sc), [[0.5, 0.5, 0.5]])
    assert np.allclose(rv.utils.calc_prob_logits_beam_search_scores(torch.tensor(sc)).numpy(), [[0.5, 0.5, 0.5]])


def test_fp32_twin_close_to_fp64(rv, oracle):
    cfg = rv.RvConfig()
    w = rv.weights.flat_to_nested(cfg, rv.weights.init_weights(cfg, seed=22))
    raw, ev, _ = rv.synthetic.make_slab(3, 40, 8, seed=5)
    t64, s64 = oracle.beam_search(w, cfg.oracle_cfg(), raw, ev, 5, 10, dtype=np.float64)
    t32, s32 = oracle.beam_search(w, cfg.oracle_cfg(), raw, ev, 5, 10, dtype=np.float32)
    assert (t64 == t32).all() and np.abs(s64 - s32).max() < 1e-4


@pytest.mark.parametrize("mode,att,depth,W", [("joint", "luong", 2, 5), ("raw", "bahdanau", 1, 3),
                                               ("event", "luong", 3, 1), ("joint", "luong", 2, 8)])
def test_c_port_matches_numpy(rv, oracle, mode, att, depth, W):
    from oracle import cpu_port
    cfg = rv.RvConfig(mode=mode, attention=att, enc_depth=depth)
    flat = rv.weights.init_weights(cfg, seed=11)
    w, blob = rv.weights.flat_to_nested(cfg, flat), rv.weights.pack(cfg, flat)
    raw, ev, _ = rv.synthetic.make_slab(9, 36, 12, seed=W, max_event_pad=4)    # 9 rows: ragged C tile (8+1)
    tok, sc = oracle.beam_search(w, cfg.oracle_cfg(), raw, ev, W, 11)
    ctok, csc = cpu_port.run(cfg.oracle_cfg(), depth, 7, blob, raw, ev, W, 11)
    assert ctok.shape == tok.shape and (ctok == tok).all()
    assert np.abs(csc - sc).max() < 1e-4
    g, lg = oracle.greedy_search(w, cfg.oracle_cfg(), raw, ev, 11)
    cg, clg = cpu_port.run(cfg.oracle_cfg(), depth, 7, blob, raw, ev, 1, 11, greedy=True)
    assert (cg == g).all() and np.abs(clg - lg).max() < 1e-4


@pytest.mark.parametrize("name", ["joint_luong_w5", "joint_bahdanau_w3", "raw_luong_greedy", "event_luong_w2_d1"])
def test_oracle_reproduces_golden(rv, oracle, name):
    """The committed fixtures regenerate from their seeds (weights + inputs are exact splitmix64
    arithmetic) and the fp32 twin + the C port land on the same tokens."""
    from oracle import cpu_port
    mg = _golden_mod()
    cfg, flat, w, raw, ev, W, L = mg.build(name)
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    assert abs(float(np.sum(rv.weights.pack(cfg, flat).astype(np.float64))) - float(g["weight_checksum"])) < 1e-9
    taps = {}
    if "greedy" in name:
        tok, logits = oracle.greedy_search(w, cfg.oracle_cfg(), raw, ev, L, taps=taps)
        assert np.abs(logits - g["logits"]).max() < 1e-5
        ctok, clg = cpu_port.run(cfg.oracle_cfg(), cfg.enc_depth, 7, rv.weights.pack(cfg, flat), raw, ev, 1, L, greedy=True)
        assert np.abs(clg - g["logits"]).max() < 1e-4
    else:
        tok, sc = oracle.beam_search(w, cfg.oracle_cfg(), raw, ev, W, L, taps=taps)
        assert np.abs(sc - g["scores"]).max() < 1e-5
        assert (taps["step_ids"] == g["step_ids"]).all() and (taps["parent_ids"] == g["parent_ids"]).all()
        ctok, csc = cpu_port.run(cfg.oracle_cfg(), cfg.enc_depth, 7, rv.weights.pack(cfg, flat), raw, ev, W, L)
        assert np.abs(csc - g["scores"]).max() < 1e-4
    assert (tok == g["tokens"]).all() and (ctok == g["tokens"]).all()
    assert (taps["mask"] == g["mask"]).all()
    assert np.abs(taps["enc_output"][:, ::5, ::16] - g["enc_output_sample"]).max() < 1e-6
    assert oracle.tokens_to_nuc_sequences(tok) == [str(x) for x in g["strings"]]


@pytest.mark.parametrize("mode,attention,W,dec_depth", [("joint", "luong", 5, 1), ("raw", "bahdanau", 3, 1), ("event", "luong", 1, 2)])
def test_torch_eager_band_matches_oracle(rv, oracle, mode, attention, W, dec_depth):
    """bench.py's second CPU engine (oracle/torch_eager.py, TF-eager-like op granularity) computes the same calls as
    the numpy oracle: tokens identical, scores within fp32 rounding."""
    from oracle import torch_eager
    cfg = rv.RvConfig(mode=mode, attention=attention, dec_depth=dec_depth)
    flat = rv.weights.init_weights(cfg, seed=6)
    flat["b_fc"][cfg.end_token] = 0.7           # early finishes: finished-beam masking and gather_tree padding
    w = rv.weights.flat_to_nested(cfg, flat)
    raw, ev, _ = rv.synthetic.make_slab(5, 40, 12, seed=2)
    tok, sc = torch_eager.beam_search(w, cfg.oracle_cfg(), raw, ev, W, 14)
    otok, osc = oracle.beam_search(w, cfg.oracle_cfg(), raw, ev, W, 14, dtype=np.float32)
    assert tok.shape == otok.shape and (tok == otok).all()
    assert np.abs(sc - osc).max() < 1e-4
