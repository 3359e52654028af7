"""torch-CPU eager restatement of the hot path -- TEST INFRASTRUCTURE / CPU BASELINE BAND, NOT PRODUCT.

SURVEY.md 8d asks for a second CPU engine "closer to TF-eager op granularity" beside the C port: the same
algorithm as oracle/ravvent_oracle.py (every function there cites the reference call site it follows), with one
torch op where TensorFlow eager runs one op -- a MatMul, a BiasAdd, a Sigmoid ... per timestep of the
`tf.while_loop`s behind Encoder.call (/root/reference/basecaller.py:48-59) and dynamic_decode (:306-313), the
attention memory tiled W times like `tile_batch` (:300-301).  fp32, torch's own intra-op thread pool.
Only bench.py's `cpu_baseline` leg and tests/ import this module.
"""
from __future__ import annotations

import numpy as np
import torch

F32_MIN = float(np.finfo(np.float32).min)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32)))


def _lstm_cell(x, h, c, W, U, b):
    """Keras LSTMCell implementation=2: z = (x.W + h.U) + b; split i, f, c~, o."""
    z = torch.add(torch.add(torch.matmul(x, W), torch.matmul(h, U)), b)
    zi, zf, zc, zo = torch.split(z, h.shape[-1], dim=-1)
    c2 = torch.add(torch.mul(torch.sigmoid(zf), c), torch.mul(torch.sigmoid(zi), torch.tanh(zc)))
    return torch.mul(torch.sigmoid(zo), torch.tanh(c2)), c2


def _bilstm(x, fwd, bwd, init):
    """Bidirectional(RNN(LSTMCell)) without a mask (basecaller.py:19-32,400,403)."""
    B, T, _ = x.shape
    u = fwd[1].shape[0]
    if init is None:
        hf = cf = hb = cb = torch.zeros((B, u))
    else:
        hf, cf, hb, cb = init
    of, ob = [], [None] * T
    for t in range(T):
        hf, cf = _lstm_cell(x[:, t], hf, cf, *fwd)
        of.append(hf)
    for t in range(T - 1, -1, -1):
        hb, cb = _lstm_cell(x[:, t], hb, cb, *bwd)
        ob[t] = hb
    return torch.cat([torch.stack(of, 1), torch.stack(ob, 1)], -1), (hf, cf, hb, cb)


def _encoder(x, layers):
    """Encoder.call (basecaller.py:48-59): layer l+1 starts from layer l's final states."""
    st = None
    for lw in layers:
        x, st = _bilstm(x, tuple(_t(a) for a in lw["fwd"]), tuple(_t(a) for a in lw["bwd"]), st)
    return x


def beam_search(weights, cfg, raw, event, beam_width, max_output_len):
    """Basecaller.beam_search_prediction (basecaller.py:296-315) -> (tokens [B,S] i32, scores [B,S] f32), numpy."""
    W = int(beam_width)
    end, start, pad = cfg["end_token"], cfg["start_token"], cfg.get("padding_value", 0.0)
    outs, masks = [], []
    with torch.no_grad():
        if cfg["mode"] in ("raw", "joint"):
            r = _t(raw); masks.append(torch.all(r != pad, -1)); outs.append(_encoder(r, weights["enc_raw"]))
        if cfg["mode"] in ("event", "joint"):
            e = _t(event); masks.append(torch.all(e != pad, -1)); outs.append(_encoder(e, weights["enc_event"]))
        enc, mask = torch.cat(outs, 1), torch.cat(masks, 1)
        B, Tm, _ = enc.shape
        N = B * W
        # tile_batch (basecaller.py:300-301) + setup_memory (:303)
        enc_t = enc.repeat_interleave(W, 0)
        mask_t = mask.repeat_interleave(W, 0)
        values = enc_t * mask_t[..., None].to(enc.dtype)
        keys = torch.matmul(values, _t(weights["W_mem"]))
        cells = [tuple(_t(a) for a in cw) for cw in weights["dec_cells"]]
        W_att, W_fc, b_fc = _t(weights["W_att"]), _t(weights["W_fc"]), _t(weights["b_fc"])
        bah = cfg["attention_type"] == "bahdanau"
        if bah:
            W_q, v_att = _t(weights["W_q"]), _t(weights["v_att"])
        V, d = W_fc.shape[1], W_att.shape[1]
        eye = torch.eye(V)
        tok = torch.full((N,), start, dtype=torch.long)
        att = torch.zeros((N, d))
        states = [(torch.zeros((N, d)), torch.zeros((N, d))) for _ in cells]
        log_probs = torch.full((B, W), -float("inf")); log_probs[:, 0] = 0.0
        finished = torch.zeros((B, W), dtype=torch.bool)
        lengths = torch.zeros((B, W), dtype=torch.long)
        fin_row = torch.full((V,), F32_MIN); fin_row[end] = 0.0
        ids, parents, scores = [], [], []
        for _ in range(int(max_output_len) - 1):
            if bool(finished.all()):
                break
            x = torch.cat([eye[tok], att], -1)                       # one_hot embedding + cell_input_fn
            new_states = []
            for (h, c), cw in zip(states, cells):
                h2, c2 = _lstm_cell(x, h, c, *cw)
                new_states.append((h2, c2)); x = h2
            q = x
            if bah:
                score = torch.sum(v_att * torch.tanh(keys + torch.matmul(q, W_q)[:, None, :]), -1)
            else:
                score = torch.matmul(keys, q[:, :, None])[:, :, 0]
            score = torch.where(mask_t, score, torch.full_like(score, -float("inf")))
            alpha = torch.softmax(score, -1)
            ctx = torch.matmul(alpha[:, None, :], values)[:, 0]
            att_new = torch.matmul(torch.cat([q, ctx], -1), W_att)
            logits = torch.add(torch.matmul(att_new, W_fc), b_fc).reshape(B, W, V)
            # _beam_search_step
            lp = torch.log_softmax(logits, -1)
            lp = torch.where(finished[..., None], fin_row, lp)
            total = (log_probs[..., None] + lp).reshape(B, W * V)
            idx = torch.argsort(-total, dim=1, stable=True)[:, :W]   # top_k, ties -> lower index
            top = torch.gather(total, 1, idx)
            word, parent = idx % V, idx // V
            prev_fin = torch.gather(finished, 1, parent)
            finished = prev_fin | (word == end)
            lengths = torch.gather(lengths, 1, parent) + (~prev_fin).long()
            log_probs = top
            flat_parent = (parent + (torch.arange(B) * W)[:, None]).reshape(-1)
            att = att_new[flat_parent]
            states = [(h[flat_parent], c[flat_parent]) for h, c in new_states]
            tok = word.reshape(-1) if not bool(finished.all()) else torch.full((N,), start, dtype=torch.long)
            ids.append(word); parents.append(parent); scores.append(top)
        if not ids:
            return np.zeros((B, 0), np.int32), np.zeros((B, 0), np.float32)
        step_ids = torch.stack(ids).numpy(); parent_ids = torch.stack(parents).numpy()
        maxlen = lengths.max(1).values.numpy()
    # gather_tree, beam 0 only (basecaller.py:315)
    S = step_ids.shape[0]
    out = np.full((B, S), end, np.int32)
    for b in range(B):
        Lb = min(S, int(maxlen[b]))
        if Lb <= 0:
            continue
        out[b, Lb - 1] = step_ids[Lb - 1, b, 0]
        p = parent_ids[Lb - 1, b, 0]
        for t in range(Lb - 2, -1, -1):
            out[b, t] = step_ids[t, b, p]; p = parent_ids[t, b, p]
        hit = np.nonzero(out[b, :Lb] == end)[0]
        if hit.size:
            out[b, hit[0]:Lb] = end
    return out, torch.stack(scores).numpy().transpose(1, 0, 2)[:, :, 0].astype(np.float32)
