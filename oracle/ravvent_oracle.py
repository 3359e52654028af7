"""CPU oracle for the Ravvent inference hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The shipped path (``ravvent-basecaller_amd``) never does; it fails
loudly when the HIP library is missing.

PARITY UNPINNED.  The arithmetic of the reference path lives in un-vendored third-party
packages (TensorFlow >= 2.7 per /root/reference/README.md:21, tensorflow_addons imported at
/root/reference/basecaller.py:3, version never stated) that are not installed in the build
container and do not exist on the GPU box.  The reference holds no tests, golden vectors or
weights for this path.  This file is therefore a restatement of the
*published* behaviour of the Keras / TFA layers at the reference's own call sites (SURVEY.md
appendix A), cross-checked layer-by-layer against ``torch.nn.LSTM`` (tests/test_oracle.py)
and against hand-computed micro cases for the beam step and gather_tree.

Every function cites the reference call site it follows.  Plain numpy; ``dtype`` selects the
arithmetic type (float64 = oracle of record, float32 = twin that mimics the reference's
compute type).
"""
from __future__ import annotations

import numpy as np

F32_MIN = float(np.finfo(np.float32).min)  # tf.float32.min, used by TFA _mask_probs


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def _sigmoid(x):
    # numerically stable logistic, same value as 1/(1+exp(-x)) to rounding
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def input_mask(x, padding_value=0.0):
    """utils.input_mask (/root/reference/utils.py:26-32): a timestep is real data iff ALL
    of its features differ from the padding value."""
    return np.all(x != padding_value, axis=-1)


def lstm_cell(x, h, c, W, U, b):
    """Keras LSTMCell, implementation=2 (cell built at
    /root/reference/basecaller.py:22-24 and :86).  Gate order i, f, c~, o;
    z = (x.W + h.U) + b."""
    u = h.shape[-1]
    z = (x @ W + h @ U) + b
    i = _sigmoid(z[..., 0 * u:1 * u])
    f = _sigmoid(z[..., 1 * u:2 * u])
    g = np.tanh(z[..., 2 * u:3 * u])
    o = _sigmoid(z[..., 3 * u:4 * u])
    c2 = f * c + i * g
    h2 = o * np.tanh(c2)
    return h2, c2


def bilstm_layer(x, fwd, bwd, init):
    """Bidirectional(RNN(LSTMCell, return_sequences, return_state)), merge 'concat'
    (/root/reference/basecaller.py:19-32).  ``fwd``/``bwd`` are (W, U, b);
    ``init`` is None (zeros) or (h_f, c_f, h_b, c_b).  No mask: padded steps run like data
    (/root/reference/basecaller.py:400,403 pass none)."""
    B, T, _ = x.shape
    u = fwd[1].shape[0]
    dt = x.dtype
    if init is None:
        hf = np.zeros((B, u), dt); cf = np.zeros((B, u), dt)
        hb = np.zeros((B, u), dt); cb = np.zeros((B, u), dt)
    else:
        hf, cf, hb, cb = [a.astype(dt) for a in init]
    out = np.empty((B, T, 2 * u), dt)
    for t in range(T):
        hf, cf = lstm_cell(x[:, t], hf, cf, *fwd)
        out[:, t, :u] = hf
    for t in range(T - 1, -1, -1):
        hb, cb = lstm_cell(x[:, t], hb, cb, *bwd)
        out[:, t, u:] = hb
    return out, (hf, cf, hb, cb)


def encoder(x, layers):
    """Encoder.call (/root/reference/basecaller.py:48-59): layer l+1 starts from layer l's
    four final states (:53,57); layer 0 from zeros."""
    states = None
    out = x
    for lw in layers:
        out, states = bilstm_layer(out, lw["fwd"], lw["bwd"], states)
    return out, states


def encode_input(weights, raw, event, mode, padding_value=0.0, dtype=np.float64):
    """Basecaller._encode_input (/root/reference/basecaller.py:395-416): mask from the
    un-encoded input; joint = raw encoder output then event encoder output on the TIME axis."""
    outs, masks = [], []
    if mode in ("raw", "joint"):
        r = np.asarray(raw, dtype)
        masks.append(input_mask(r, padding_value))
        outs.append(encoder(r, _cast_layers(weights["enc_raw"], dtype))[0])
    if mode in ("event", "joint"):
        e = np.asarray(event, dtype)
        masks.append(input_mask(e, padding_value))
        outs.append(encoder(e, _cast_layers(weights["enc_event"], dtype))[0])
    return np.concatenate(outs, axis=1), np.concatenate(masks, axis=1)


def _cast_layers(layers, dtype):
    return [{d: tuple(np.asarray(a, dtype) for a in lw[d]) for d in ("fwd", "bwd")} for lw in layers]


def setup_memory(weights, enc_output, mask, dtype):
    """attention_mechanism.setup_memory (/root/reference/basecaller.py:303):
    values = memory * mask ; keys = values . W_mem (Dense, no bias)."""
    values = enc_output * mask[..., None].astype(dtype)
    keys = values @ np.asarray(weights["W_mem"], dtype)
    return keys, values


def attention_step(weights, tok, att_prev, cell_states, keys, values, mask, attention_type, dtype):
    """One AttentionWrapper step + output layer (wiring at
    /root/reference/basecaller.py:83-94,119-122).  Rows are [..., ] leading dims shared by all
    state tensors; keys/values/mask broadcast against them (the reference tiles them W times
    with tile_batch, :300-301, which is the same arithmetic).

    returns logits, att, alpha, new_cell_states"""
    V = np.asarray(weights["W_fc"]).shape[1]
    onehot = np.eye(V, dtype=dtype)[tok]                      # Decoder.embedding (:83)
    x = np.concatenate([onehot, att_prev], axis=-1)           # cell_input_fn = concat
    new_states = []
    for (h, c), cw in zip(cell_states, weights["dec_cells"]):  # StackedRNNCells (:85-91)
        h2, c2 = lstm_cell(x, h, c, *(np.asarray(a, dtype) for a in cw))
        new_states.append((h2, c2))
        x = h2
    q = x
    if attention_type == "luong":      # LuongAttention(scale=False): q . keys
        score = np.einsum("...d,...td->...t", q, keys)
    elif attention_type == "bahdanau":  # BahdanauAttention(normalize=False)
        pq = q @ np.asarray(weights["W_q"], dtype)
        score = np.sum(np.asarray(weights["v_att"], dtype) * np.tanh(keys + pq[..., None, :]), axis=-1)
    else:
        raise ValueError(attention_type)
    score = np.where(mask, score, -np.inf)                     # _maybe_mask_score
    m = np.max(score, axis=-1, keepdims=True)
    e = np.exp(score - m)
    alpha = e / np.sum(e, axis=-1, keepdims=True)              # softmax over T_m
    ctx = np.einsum("...t,...te->...e", alpha, values)
    att = np.concatenate([q, ctx], axis=-1) @ np.asarray(weights["W_att"], dtype)  # no bias/act
    logits = att @ np.asarray(weights["W_fc"], dtype) + np.asarray(weights["b_fc"], dtype)  # fc (:94)
    return logits, att, alpha, new_states


def log_softmax(x):
    m = np.max(x, axis=-1, keepdims=True)
    s = x - m
    return s - np.log(np.sum(np.exp(s), axis=-1, keepdims=True))


def gather_tree(step_ids, parent_ids, max_sequence_lengths, end_token):
    """TFA gather_tree, reached through BeamSearchDecoder.finalize from
    /root/reference/basecaller.py:313.  step_ids/parent_ids are time-major [S,B,W]."""
    S, B, W = step_ids.shape
    out = np.full_like(step_ids, end_token)
    for b in range(B):
        L = min(S, int(max_sequence_lengths[b]))
        if L <= 0:
            continue
        for w in range(W):
            out[L - 1, b, w] = step_ids[L - 1, b, w]
            p = parent_ids[L - 1, b, w]
            for t in range(L - 2, -1, -1):
                out[t, b, w] = step_ids[t, b, p]
                p = parent_ids[t, b, p]
            done = False
            for t in range(L):
                if done:
                    out[t, b, w] = end_token
                elif out[t, b, w] == end_token:
                    done = True
    return out


def beam_search_step(logits, log_probs, finished, lengths, end_token):
    """TFA _beam_search_step with length/coverage penalty 0.
    logits [B,W,V]; returns (scores[B,W], word[B,W], parent[B,W], log_probs', finished',
    lengths')."""
    B, W, V = logits.shape
    lp = log_softmax(logits)
    fin_row = np.full((V,), F32_MIN, lp.dtype)
    fin_row[end_token] = 0.0
    lp = np.where(finished[..., None], fin_row, lp)            # _mask_probs
    total = log_probs[..., None] + lp
    flat = total.reshape(B, W * V)
    # tf.math.top_k: descending, ties -> lower index first
    idx = np.argsort(-flat, axis=1, kind="stable")[:, :W]
    top = np.take_along_axis(flat, idx, axis=1)
    word = (idx % V).astype(np.int32)
    parent = (idx // V).astype(np.int32)
    prev_fin = np.take_along_axis(finished, parent, axis=1)
    new_fin = prev_fin | (word == end_token)
    new_len = np.take_along_axis(lengths, parent, axis=1) + (~prev_fin).astype(np.int64)
    return top, word, parent, top.copy(), new_fin, new_len


def beam_search(weights, cfg, raw, event, beam_width, max_output_len, dtype=np.float64, taps=None):
    """Basecaller.beam_search_prediction (/root/reference/basecaller.py:296-315).

    cfg: dict(mode, attention_type, start_token, end_token, padding_value).
    returns tokens [B,S] int32, scores [B,S] dtype.  ``taps`` (dict) receives enc_output, mask,
    keys, step_logits [S,B,W,V], step_alignments [S,B,W,Tm], step_ids, parent_ids."""
    W = int(beam_width)
    end, start = cfg["end_token"], cfg["start_token"]
    enc_out, mask = encode_input(weights, raw, event, cfg["mode"], cfg.get("padding_value", 0.0), dtype)
    keys, values = setup_memory(weights, enc_out, mask, dtype)
    B = enc_out.shape[0]
    d = np.asarray(weights["W_att"]).shape[1]
    depth = len(weights["dec_cells"])
    keys_b, values_b, mask_b = keys[:, None], values[:, None], mask[:, None]  # broadcast over W == tile_batch

    tok = np.full((B, W), start, np.int32)
    att = np.zeros((B, W, d), dtype)
    states = [(np.zeros((B, W, d), dtype), np.zeros((B, W, d), dtype)) for _ in range(depth)]
    log_probs = np.full((B, W), -np.inf, dtype)
    log_probs[:, 0] = 0.0
    finished = np.zeros((B, W), bool)
    lengths = np.zeros((B, W), np.int64)

    ids, parents, scores, all_logits, all_align = [], [], [], [], []
    max_iter = int(max_output_len) - 1
    for _ in range(max_iter):
        if finished.all():
            break
        logits, att_new, alpha, new_states = attention_step(
            weights, tok, att, states, keys_b, values_b, mask_b, cfg["attention_type"], dtype)
        top, word, parent, log_probs, finished, lengths = beam_search_step(
            logits, log_probs, finished, lengths, end)
        g = lambda a: np.take_along_axis(a, parent[..., None], axis=1)
        att = g(att_new)
        states = [(g(h), g(c)) for (h, c) in new_states]
        tok = word if not finished.all() else np.full((B, W), start, np.int32)
        ids.append(word); parents.append(parent); scores.append(top)
        all_logits.append(logits); all_align.append(alpha)
    S = len(ids)
    if S == 0:
        return np.zeros((B, 0), np.int32), np.zeros((B, 0), dtype)
    step_ids = np.stack(ids); parent_ids = np.stack(parents)
    pred = gather_tree(step_ids, parent_ids, lengths.max(axis=1).astype(np.int32), end)
    if taps is not None:
        taps.update(enc_output=enc_out, mask=mask, keys=keys, step_logits=np.stack(all_logits),
                    step_alignments=np.stack(all_align), step_ids=step_ids, parent_ids=parent_ids,
                    lengths=lengths, finished=finished)
    tokens = np.transpose(pred, (1, 0, 2))[:, :, 0].astype(np.int32)
    top1 = np.transpose(np.stack(scores), (1, 0, 2))[:, :, 0]
    return tokens, top1


def greedy_search(weights, cfg, raw, event, max_output_len, dtype=np.float64, taps=None):
    """Basecaller.greedy_search_prediction (/root/reference/basecaller.py:317-330):
    BasicDecoder + GreedyEmbeddingSampler, impute_finished=False.
    returns sample_id [B,S] int32, logits [B,S,V]."""
    end, start = cfg["end_token"], cfg["start_token"]
    enc_out, mask = encode_input(weights, raw, event, cfg["mode"], cfg.get("padding_value", 0.0), dtype)
    keys, values = setup_memory(weights, enc_out, mask, dtype)
    B = enc_out.shape[0]
    d = np.asarray(weights["W_att"]).shape[1]
    depth = len(weights["dec_cells"])
    tok = np.full((B,), start, np.int32)
    att = np.zeros((B, d), dtype)
    states = [(np.zeros((B, d), dtype), np.zeros((B, d), dtype)) for _ in range(depth)]
    finished = np.zeros((B,), bool)
    ids, all_logits, all_align = [], [], []
    for _ in range(int(max_output_len) - 1):
        if finished.all():
            break
        logits, att, alpha, states = attention_step(
            weights, tok, att, states, keys, values, mask, cfg["attention_type"], dtype)
        sample = np.argmax(logits, axis=-1).astype(np.int32)   # first max on ties
        finished = finished | (sample == end)
        tok = sample
        ids.append(sample); all_logits.append(logits); all_align.append(alpha)
    if taps is not None:
        taps.update(enc_output=enc_out, mask=mask, keys=keys,
                    step_alignments=np.stack(all_align) if all_align else None)
    if not ids:
        V = np.asarray(weights["W_fc"]).shape[1]
        return np.zeros((B, 0), np.int32), np.zeros((B, 0, V), dtype)
    return np.stack(ids, axis=1), np.stack(all_logits, axis=1)


INDEX_WORD = {0: "", 1: "^", 2: "$", 3: "a", 4: "c", 5: "g", 6: "t"}  # data_loader.py:20-22


def tokens_to_nuc_sequences(tokens):
    """Basecaller.tokens_to_nuc_sequences (/root/reference/basecaller.py:289-294)."""
    out = []
    for row in np.asarray(tokens):
        text = " ".join(INDEX_WORD[int(t)] for t in row if int(t) in INDEX_WORD)
        out.append(text.replace(" ", "").replace("^", "").replace("$", "").upper())
    return out


def calc_prob_logits_beam_search_scores(scores):
    """utils.calc_prob_logits_beam_search_scores (/root/reference/utils.py:123-128)."""
    s = np.asarray(scores)
    prev = np.zeros_like(s)
    prev[..., 1:] = s[..., :-1]
    return np.exp(s - prev)
