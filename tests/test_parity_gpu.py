"""GPU parity tests proper: the HIP path, called through the C-ABI (ctypes) behind the
`Basecaller` API, against the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): base-call strings bit-exact at beam=1, attention /
logit tensors within 1e-4 (fp32) of the fp64 oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _mk(rv, mode="joint", attention="luong", enc_depth=2, seed=22, gain=1.0, **kw):
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, mode, 0.0, encoder_depth=enc_depth,
                       attention_type=attention, honor_attention_type=True, max_batch=kw.pop("max_batch", 64),
                       **kw)
    flat = bc.init_random_weights(seed=seed, gain=gain)
    return bc, rv.weights.flat_to_nested(bc.cfg, flat)


def _inputs(rv, mode, raw, ev):
    return {"joint": (raw, ev), "raw": raw, "event": ev}[mode]


@pytest.mark.parametrize("mode,attention,enc_depth", [
    ("joint", "luong", 2), ("raw", "luong", 2), ("event", "luong", 2),
    ("joint", "bahdanau", 2), ("joint", "luong", 1), ("joint", "luong", 3)])
def test_greedy_tensors_and_strings(rv, oracle, mode, attention, enc_depth):
    bc, w = _mk(rv, mode, attention, enc_depth)
    bc.set_option("debug_taps", 1)
    raw, ev, _ = rv.synthetic.make_slab(5, 60, 12, seed=3)
    L = 14
    tok, logits = bc.greedy_search_prediction(_inputs(rv, mode, raw, ev), L)
    taps = {}
    otok, ologits = oracle.greedy_search(w, bc.cfg.oracle_cfg(), raw, ev, L, dtype=np.float64, taps=taps)
    S = otok.shape[1]
    assert tok.shape == (5, S) and logits.shape == (5, S, 7)
    B, Tm = taps["mask"].shape
    enc = bc.get_tensor("enc_output").reshape(B, Tm, 256)
    assert np.abs(enc - taps["enc_output"]).max() < TOL
    assert (bc.get_tensor("mask").reshape(B, Tm) == taps["mask"]).all()
    assert np.abs(bc.get_tensor("keys").reshape(B, Tm, 128) - taps["keys"]).max() < TOL
    al = bc.get_tensor("step_alignments").reshape(S, B, 1, Tm)[:, :, 0]
    assert np.abs(al - taps["step_alignments"]).max() < TOL
    assert np.abs(logits.numpy() - ologits).max() < TOL
    assert (tok.numpy() == otok).all()                      # beam=1 strings bit-exact
    assert bc.tokens_to_nuc_sequences(tok) == oracle.tokens_to_nuc_sequences(otok)
    bc.close()


@pytest.mark.parametrize("W", [1, 3, 5, 8])
def test_beam_search_matches_oracle(rv, oracle, W):
    bc, w = _mk(rv)
    bc.set_option("debug_taps", 1)
    raw, ev, _ = rv.synthetic.make_slab(6, 50, 10, seed=W)
    L = 16
    tok, sc = bc.beam_search_prediction((raw, ev), beam_width=W, max_output_len=L)
    taps = {}
    otok, osc = oracle.beam_search(w, bc.cfg.oracle_cfg(), raw, ev, W, L, dtype=np.float64, taps=taps)
    assert tok.shape == otok.shape
    S = otok.shape[1]
    lg = bc.get_tensor("step_logits").reshape(S, 6, W, 7)
    # logits of live beams (dead -inf beams at step 0 carry identical state in both)
    assert np.abs(lg - taps["step_logits"]).max() < TOL
    assert (bc.get_tensor("step_ids").reshape(S, 6, W) == taps["step_ids"]).all()
    assert (bc.get_tensor("parent_ids").reshape(S, 6, W) == taps["parent_ids"]).all()
    assert (tok.numpy() == otok).all()
    assert np.abs(sc.numpy() - osc).max() < TOL
    bc.close()
