"""Weight container for the Ravvent inference path: structure, initialisers, flat blob.

No checkpoint ships with the reference (`models/.gitkeep` only) and Keras `load_weights`
(/root/reference/ravvent_performance_evaluator.py:107) reads TF-format checkpoints that need
TensorFlow; this build defines its own flat little-endian fp32 blob (order below) plus an
``.npz`` file form.  Parameter inventory follows SURVEY.md A.7.

Blob order (all matrices row-major ``[in, out]`` exactly as Keras stores kernels):
  for enc in (raw, event): for layer: for dir in (fwd, bwd): W[F,4u]  U[u,4u]  b[4u]
  for k in decoder cells: Wd[in_k,4d]  Ud[d,4d]  bd[4d]      (in_0 = vocab + d, in_k = d)
  W_mem[2u,d]   W_q[d,d]  v_att[d]  (both always present; used only by Bahdanau)
  W_att[d+2u,d] W_fc[d,V] b_fc[V]
"""
from __future__ import annotations

import numpy as np

from .config import RvConfig, RAW_FEATURES, EVENT_FEATURES


def _enc_in_features(cfg: RvConfig, enc: str, layer: int) -> int:
    if layer > 0:
        return 2 * cfg.enc_units
    return RAW_FEATURES if enc == "raw" else EVENT_FEATURES


def blob_layout(cfg: RvConfig):
    """[(name, shape)] in blob order."""
    u, d, V = cfg.enc_units, cfg.dec_units, cfg.vocab
    out = []
    for enc in ("raw", "event"):
        for l in range(cfg.enc_depth):
            F = _enc_in_features(cfg, enc, l)
            for dr in ("fwd", "bwd"):
                p = f"enc_{enc}.{l}.{dr}"
                out += [(p + ".W", (F, 4 * u)), (p + ".U", (u, 4 * u)), (p + ".b", (4 * u,))]
    for k in range(cfg.dec_depth):
        fin = V + d if k == 0 else d
        p = f"dec_cells.{k}"
        out += [(p + ".W", (fin, 4 * d)), (p + ".U", (d, 4 * d)), (p + ".b", (4 * d,))]
    out += [("W_mem", (2 * u, d)), ("W_q", (d, d)), ("v_att", (d,)),
            ("W_att", (d + 2 * u, d)), ("W_fc", (d, V)), ("b_fc", (V,))]
    return out


def blob_size(cfg: RvConfig) -> int:
    return int(sum(int(np.prod(s)) for _, s in blob_layout(cfg)))


# ------------------------------------------------------------------ flat <-> nested
def flat_to_nested(cfg: RvConfig, flat: dict) -> dict:
    """name->array dict into the nested form the CPU oracle takes."""
    w = {"enc_raw": [], "enc_event": [], "dec_cells": []}
    for enc in ("raw", "event"):
        for l in range(cfg.enc_depth):
            w[f"enc_{enc}"].append({dr: tuple(flat[f"enc_{enc}.{l}.{dr}.{n}"] for n in "WUb")
                                     for dr in ("fwd", "bwd")})
    for k in range(cfg.dec_depth):
        w["dec_cells"].append(tuple(flat[f"dec_cells.{k}.{n}"] for n in "WUb"))
    for n in ("W_mem", "W_q", "v_att", "W_att", "W_fc", "b_fc"):
        w[n] = flat[n]
    return w


def pack(cfg: RvConfig, flat: dict) -> np.ndarray:
    parts = []
    for name, shape in blob_layout(cfg):
        a = np.asarray(flat[name], np.float32)
        if a.shape != tuple(shape):
            raise ValueError(f"weight {name}: shape {a.shape}, expected {shape}")
        parts.append(a.ravel())
    return np.ascontiguousarray(np.concatenate(parts), dtype="<f4")


def unpack(cfg: RvConfig, blob: np.ndarray) -> dict:
    blob = np.asarray(blob, np.float32).ravel()
    if blob.size != blob_size(cfg):
        raise ValueError(f"blob has {blob.size} floats, config needs {blob_size(cfg)}")
    flat, off = {}, 0
    for name, shape in blob_layout(cfg):
        n = int(np.prod(shape))
        flat[name] = blob[off:off + n].reshape(shape).copy()
        off += n
    return flat


def save(path: str, cfg: RvConfig, flat: dict) -> None:
    np.savez(path, **{k: np.asarray(v, np.float32) for k, v in flat.items()})


def load(path: str, cfg: RvConfig) -> dict:
    with np.load(path if str(path).endswith(".npz") else str(path) + ".npz") as z:
        flat = {k: z[k] for k in z.files}
    missing = [n for n, _ in blob_layout(cfg) if n not in flat]
    if missing:
        raise KeyError(f"weight file {path} lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
    return flat


# ------------------------------------------------------------------ initialisers
def _glorot(rng, shape):
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, shape)


def _orthogonal(rng, shape):
    a = rng.standard_normal((max(shape), min(shape)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    return (q if shape[0] >= shape[1] else q.T)[:shape[0], :shape[1]]


def _splitmix_uniform(seed: int, n: int) -> np.ndarray:
    """n exactly reproducible uniforms in [-1, 1): splitmix64 in uint64 arithmetic (no BLAS,
    no LAPACK, no generator state), so golden fixtures regenerate bit-identically anywhere."""
    with np.errstate(over="ignore"):
        x = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64(seed) * np.uint64(0xD1342543DE82EF95))
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def init_weights(cfg: RvConfig, seed: int = 22, scheme: str = "keras", gain: float = 1.0) -> dict:
    """Random-init weights of the reference architecture (flat name->fp32 array).

    scheme 'keras': the Keras defaults the reference relies on -- Glorot-uniform kernels
      (/root/reference/basecaller.py:23,86), orthogonal recurrent kernels, zero bias with
      unit forget-gate bias; default seed 22 as /root/reference/ravvent.py:9.
    scheme 'hash' : splitmix64 uniforms scaled like Glorot; exactly reproducible, used for the
      committed golden fixtures.
    ``gain`` scales the attention/output matrices ("peaky" variant: larger logit margins).
    """
    flat = {}
    rng = np.random.default_rng(seed)
    for i, (name, shape) in enumerate(blob_layout(cfg)):
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "b" and name != "b_fc":
            b = np.zeros(shape)
            q = shape[0] // 4
            b[q:2 * q] = 1.0                                   # unit_forget_bias
            a = b
        elif name == "b_fc":
            a = np.zeros(shape)
        elif scheme == "hash":
            fan = shape[0] + (shape[1] if len(shape) > 1 else shape[0])
            a = _splitmix_uniform(seed * 1000 + i, int(np.prod(shape))).reshape(shape) * np.sqrt(6.0 / fan)
        elif leaf == "U":
            a = _orthogonal(rng, shape)
        elif len(shape) == 1:                                  # v_att: glorot_uniform on a vector
            lim = np.sqrt(6.0 / (shape[0] + 1))
            a = rng.uniform(-lim, lim, shape)
        else:
            a = _glorot(rng, shape)
        if name in ("W_att", "W_fc", "W_mem"):
            a = a * gain
        flat[name] = np.asarray(a, np.float32)
    return flat


def base_calling_weights(cfg: RvConfig, seed: int = 1, gain: float = 3.0, token_gain: float = 6.0, recurrent_gain: float = 6.0,
                         attention_gain: float = 4.0, forget_bias: float = -8.0, base_bias: float = 1.5,
                         other_bias: float = -60.0) -> dict:
    """Random weights whose best hypothesis is a full-length, VARIED base string.  The output bias favours a/c/g/t and pushes the
    end, start and pad tokens out of reach, so every call has max_output_len - 1 letters and the read-level merger
    (/root/reference/merger.py:155-248: 25-letter overlaps) really aligns, splices and appends.  The first decoder cell forgets
    (negative forget bias) and its token rows, recurrent kernel and attention-input rows are scaled up: the cell is then a
    chaotic map of (previous letter, attention, h), the letters change at ~3 of 4 steps, all four are used, and the calls differ
    from chunk to chunk.  (Keras-default weights settle into homopolymer runs, on which the merger's local alignment finds nothing
    and returns early, merger.py:181-197.)  Untrained weights cannot call a read CORRECTLY -- and a chaotic cell amplifies fp32
    rounding, so these weights serve comparisons of the product with ITSELF (sharded == single-GPU == merged, byte for byte), not
    comparisons against the oracle."""
    flat = init_weights(cfg, seed=seed, gain=gain)
    V, u = cfg.vocab, cfg.dec_units
    flat["dec_cells.0.W"][:V] *= token_gain
    flat["dec_cells.0.W"][V:] *= attention_gain
    flat["dec_cells.0.U"] *= recurrent_gain
    flat["dec_cells.0.b"][u:2 * u] = forget_bias
    b = flat["b_fc"]
    b[:] = base_bias
    for t in (cfg.end_token, cfg.start_token, cfg.pad_token):
        b[t] = other_bias
    return flat
