"""TF-format checkpoint -> weight blob.

The reference restores weights with Keras `load_weights(prefix)` on a TF-format checkpoint
(/root/reference/ravvent_performance_evaluator.py:107; written by `ModelCheckpoint(save_weights_only=True)`,
/root/reference/ravvent.py:61-70): a *tensor bundle* `prefix.index` + `prefix.data-00000-of-00001`.  TensorFlow is not
installed here and no checkpoint ships with the reference, so this module reads the bundle format itself (pure numpy /
struct, no TensorFlow):

  * `prefix.index` is a LevelDB-format table (sorted string table: data blocks of prefix-compressed key/value entries with
    a restart array, an index block, a 48-byte footer ending in the magic 0xdb4775248b80fb57; every block is followed by a
    1-byte compression tag and a masked CRC32C).  TensorFlow writes it uncompressed.  Key "" holds a BundleHeaderProto,
    every other key a BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}.
  * `prefix.data-SSSSS-of-NNNNN` holds the raw little-endian tensor bytes at [offset, offset + size).

Keys of an object-based Keras checkpoint are the attribute paths of the model's variables followed by
`/.ATTRIBUTES/VARIABLE_VALUE`.  `variable_paths(cfg)` lists, for every segment of this build's weight blob (weights.py),
the path the reference's classes give it (basecaller.py:7-46 Encoder.rnn_layers -> Bidirectional.forward_layer /
backward_layer -> RNN.cell -> LSTMCell.kernel / recurrent_kernel / bias; :85-94 Decoder.decoder_rnn_cell -> cells,
Decoder.fc; :110-122 attention_mechanism.memory_layer (+ query_layer, attention_v for Bahdanau), AttentionWrapper's
attention layer); `weights_manifest.json` is that table with offsets.  Those paths are derived from the class definitions,
not from a real checkpoint (none exists in the tree): `flat_from_checkpoint` therefore matches by the path's distinctive
components (encoder name, layer index, direction, leaf) rather than by the exact string, checks every shape, and lists
what it could not place.  `write_tensor_bundle` writes the same format (used by the tests, and by anyone who wants to
hand weights to TensorFlow).
"""
from __future__ import annotations

import json
import os
import re
import struct

import numpy as np

from . import weights as _weights
from .config import RvConfig

TABLE_MAGIC = 0xDB4775248B80FB57
_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8"), 10: np.dtype("bool"),
           4: np.dtype("u1"), 6: np.dtype("i1"), 5: np.dtype("<i2"), 19: np.dtype("<f2")}
_DTYPE_IDS = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9}


# ------------------------------------------------------------------ CRC32C (Castagnoli), masked as LevelDB does
def _crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC = _crc_table()


def _crc_bytes(data, state: int) -> int:
    for b in data:
        state = _CRC[(state ^ b) & 0xFF] ^ (state >> 8)
    return state


def _zero_shift_matrix(nbytes: int):
    """The CRC register after `nbytes` zero bytes is a GF(2)-linear map of the register before them: its 32 columns."""
    def mul(a, b):                                   # columns of a . b
        return [_apply(a, col) for col in b]
    m = [_crc_bytes(b"\x00", 1 << j) for j in range(32)]
    out, e = [1 << j for j in range(32)], nbytes
    while e:
        if e & 1:
            out = mul(m, out)
        m = mul(m, m)
        e >>= 1
    return out


def _apply(mat, x: int) -> int:
    r, j = 0, 0
    while x:
        if x & 1:
            r ^= mat[j]
        x >>= 1; j += 1
    return r


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC32C of `data`.  Large buffers (a checkpoint's kernels are megabytes) are cut into 1024 equal chunks whose registers advance
    together, one numpy table lookup per byte POSITION, and are then chained through the zero-shift matrix of the chunk length --
    the byte-at-a-time Python loop of the first version took seconds per checkpoint."""
    n = len(data)
    state = crc ^ 0xFFFFFFFF
    if n < 1 << 14:
        return _crc_bytes(data, state) ^ 0xFFFFFFFF
    nchunk = 1024
    L = n // nchunk
    a = np.frombuffer(data, np.uint8, count=nchunk * L).reshape(nchunk, L)
    tab = np.asarray(_CRC, np.uint32)
    c = np.zeros(nchunk, np.uint32)
    for i in range(L):
        c = tab[(c ^ a[:, i]) & 0xFF] ^ (c >> 8)
    shift = _zero_shift_matrix(L)
    for k in range(nchunk):                          # register after chunk k = shift(register before it) xor (chunk k from a zero register)
        state = _apply(shift, state) ^ int(c[k])
    return _crc_bytes(data[nchunk * L:], state) ^ 0xFFFFFFFF


def _mask(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------ varints / minimal protobuf
def _get_varint(buf: bytes, pos: int):
    out = shift = 0
    while True:
        b = buf[pos]; pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf: bytes) -> dict:
    """field number -> list of raw values (varint ints or length-delimited bytes)."""
    out, pos = {}, 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _get_varint(buf, pos)
        elif wire == 2:
            n, pos = _get_varint(buf, pos)
            v = buf[pos:pos + n]; pos += n
        elif wire == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        elif wire == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        else:
            raise ValueError(f"unsupported protobuf wire type {wire}")
        out.setdefault(field, []).append(v)
    return out


def _field(num: int, value) -> bytes:
    if isinstance(value, (bytes, bytearray)):
        return _put_varint(num << 3 | 2) + _put_varint(len(value)) + bytes(value)
    if isinstance(value, tuple) and value[0] == "fixed32":
        return _put_varint(num << 3 | 5) + struct.pack("<I", value[1])
    return _put_varint(num << 3 | 0) + _put_varint(int(value))


# ------------------------------------------------------------------ LevelDB table
def _read_block(data: bytes, offset: int, size: int, verify: bool) -> bytes:
    body, tag = data[offset:offset + size], data[offset + size]
    if verify:
        want = struct.unpack_from("<I", data, offset + size + 1)[0]
        if _mask(crc32c(data[offset:offset + size + 1])) != want:
            raise ValueError(f"index block at {offset}: CRC32C mismatch")
    if tag != 0:
        raise ValueError("compressed index block (snappy): TensorFlow writes checkpoint indices uncompressed; "
                         "re-save the checkpoint or decompress the table first")
    return body


def _block_entries(block: bytes):
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]; pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path: str, verify_crc: bool = True) -> dict:
    """key bytes -> value bytes of a LevelDB-format table file."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: not a TensorFlow checkpoint index (bad table magic)")
    footer = data[-48:]
    pos = 0
    _, pos = _get_varint(footer, pos); _, pos = _get_varint(footer, pos)        # metaindex handle
    ioff, pos = _get_varint(footer, pos); isz, pos = _get_varint(footer, pos)   # index handle
    out = {}
    for _, handle in _block_entries(_read_block(data, ioff, isz, verify_crc)):
        boff, p = _get_varint(handle, 0)
        bsz, _ = _get_varint(handle, p)
        for k, v in _block_entries(_read_block(data, boff, bsz, verify_crc)):
            out[k] = v
    return out


def _build_block(entries, restart_interval: int = 16) -> bytes:
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path: str, items: dict, block_entries: int = 8) -> None:
    """Write key -> value (bytes) as an uncompressed LevelDB-format table."""
    keys = sorted(items)
    out = bytearray()

    def emit(block: bytes):
        off = len(out)
        out.extend(block + b"\x00")
        out.extend(struct.pack("<I", _mask(crc32c(block + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block))
    index = []
    for i in range(0, len(keys), block_entries):
        part = keys[i:i + block_entries]
        index.append((part[-1], emit(_build_block([(k, items[k]) for k in part]))))
    if not index:
        index.append((b"", emit(_build_block([]))))
    meta = emit(_build_block([]))
    idx = emit(_build_block(index, restart_interval=1))
    footer = meta + idx
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(bytes(out))


# ------------------------------------------------------------------ tensor bundle
def read_tensor_bundle(prefix: str, verify_crc: bool = True, want=None) -> dict:
    """`prefix.index` + `prefix.data-*` -> {key: ndarray}.  String tensors (the object graph) are skipped; `want(key) -> bool`
    selects the tensors that are read and CRC-checked at all (a ModelCheckpoint file also carries the Adam slots of every
    variable: twice the model again, which load_weights has no use for)."""
    table = read_table(str(prefix) + ".index", verify_crc)
    header = _parse_proto(table.get(b"", b""))
    num_shards = header.get(1, [1])[0]
    if header.get(2, [0])[0] != 0:
        raise ValueError("big-endian tensor bundle")
    shards, out = {}, {}
    for key, val in table.items():
        if key == b"":
            continue
        e = _parse_proto(val)
        dtype = e.get(1, [0])[0]
        if dtype not in _DTYPES:
            continue                                    # DT_STRING etc.: nothing this build needs
        if want is not None and not want(key.decode()):
            continue
        dims = [_parse_proto(d).get(1, [0])[0] for d in _parse_proto(e.get(2, [b""])[0]).get(2, [])]
        shard, off, size = e.get(3, [0])[0], e.get(4, [0])[0], e.get(5, [0])[0]
        if 7 in e:
            raise ValueError(f"{key.decode()}: sliced (partitioned) variables are not supported")
        if shard not in shards:
            with open(f"{prefix}.data-{shard:05d}-of-{num_shards:05d}", "rb") as f:
                shards[shard] = f.read()
        raw = shards[shard][off:off + size]
        if verify_crc and 6 in e and _mask(crc32c(raw)) != e[6][0]:
            raise ValueError(f"{key.decode()}: tensor bytes fail their CRC32C")
        out[key.decode()] = np.frombuffer(raw, _DTYPES[dtype]).reshape(dims).copy()
    return out


def write_tensor_bundle(prefix: str, tensors: dict) -> None:
    """{key: ndarray} -> `prefix.index` + `prefix.data-00000-of-00001` in TensorFlow's tensor-bundle format."""
    os.makedirs(os.path.dirname(os.path.abspath(str(prefix))) or ".", exist_ok=True)
    items, data = {}, bytearray()
    items[b""] = _field(1, 1) + _field(2, 0) + _field(3, _field(1, 1))          # num_shards 1, little endian, version.producer 1
    for key in sorted(tensors):
        a = np.asarray(tensors[key], order="C")      # (ascontiguousarray would turn a 0-d counter into shape (1,))
        if a.dtype not in _DTYPE_IDS:
            raise ValueError(f"{key}: dtype {a.dtype} not supported")
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        shape = b"".join(_field(2, _field(1, int(d))) for d in a.shape)
        items[key.encode()] = (_field(1, _DTYPE_IDS[a.dtype]) + _field(2, shape) + _field(3, 0) + _field(4, len(data))
                               + _field(5, len(raw)) + _field(6, ("fixed32", _mask(crc32c(raw)))))
        data += raw
    with open(f"{prefix}.data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    write_table(str(prefix) + ".index", items)


# ------------------------------------------------------------------ blob segment <-> Keras variable path
def variable_paths(cfg: RvConfig) -> dict:
    """blob segment name (weights.blob_layout) -> the variable's attribute path in the reference's model."""
    out = {}
    leaf = {"W": "kernel", "U": "recurrent_kernel", "b": "bias"}
    for enc in ("raw", "event"):
        for l in range(cfg.enc_depth):
            for dr, lay in (("fwd", "forward_layer"), ("bwd", "backward_layer")):
                for n, kn in leaf.items():
                    out[f"enc_{enc}.{l}.{dr}.{n}"] = f"encoder_{enc}/rnn_layers/{l}/{lay}/cell/{kn}"
    for k in range(cfg.dec_depth):
        for n, kn in leaf.items():
            out[f"dec_cells.{k}.{n}"] = f"decoder/decoder_rnn_cell/cells/{k}/{kn}"
    out["W_mem"] = "decoder/attention_mechanism/memory_layer/kernel"
    out["W_q"] = "decoder/attention_mechanism/query_layer/kernel"            # Bahdanau only
    out["v_att"] = "decoder/attention_mechanism/attention_v"                  # Bahdanau only
    out["W_att"] = "decoder/rnn_cell/_attention_layers/0/kernel"
    out["W_fc"] = "decoder/fc/kernel"
    out["b_fc"] = "decoder/fc/bias"
    return out


def weights_manifest(cfg: RvConfig) -> list:
    """One record per blob segment: name, shape, float offset, checkpoint key, and what the tensor is."""
    paths, off, out = variable_paths(cfg), 0, []
    for name, shape in _weights.blob_layout(cfg):
        n = int(np.prod(shape))
        out.append({"segment": name, "shape": list(shape), "offset_floats": off, "count": n,
                    "checkpoint_key": paths[name] + _SUFFIX,
                    "optional": name in ("W_q", "v_att") and cfg.attention != "bahdanau"})
        off += n
    return out


def _classify(key: str):
    """checkpoint key -> blob segment name, or None (optimizer slots, counters, object graph ...)."""
    if not key.endswith(_SUFFIX) or ".OPTIMIZER_SLOT" in key or key.startswith("optimizer"):
        return None
    path = key[:-len(_SUFFIX)]
    leaf = path.rsplit("/", 1)[-1]
    m = re.search(r"encoder_(raw|event)\b.*?(?:rnn_layers|layer_with_weights)[/-](\d+)/(forward_layer|backward_layer)/", path)
    if m and leaf in ("kernel", "recurrent_kernel", "bias"):
        return f"enc_{m.group(1)}.{m.group(2)}.{'fwd' if m.group(3) == 'forward_layer' else 'bwd'}." + \
               {"kernel": "W", "recurrent_kernel": "U", "bias": "b"}[leaf]
    if "decoder" not in path:
        return None
    m = re.search(r"cells/(\d+)/(kernel|recurrent_kernel|bias)$", path)
    if m:
        return f"dec_cells.{m.group(1)}." + {"kernel": "W", "recurrent_kernel": "U", "bias": "b"}[m.group(2)]
    if path.endswith("memory_layer/kernel"):
        return "W_mem"
    if path.endswith("query_layer/kernel"):
        return "W_q"
    if leaf == "attention_v":
        return "v_att"
    if re.search(r"attention_layers?/(0/)?kernel$", path):
        return "W_att"
    if path.endswith("fc/kernel"):
        return "W_fc"
    if path.endswith("fc/bias"):
        return "b_fc"
    return None


def flat_from_tensors(tensors: dict, cfg: RvConfig) -> dict:
    """{checkpoint key: array} -> the flat name -> fp32 array dict `Basecaller.set_weights_flat` takes."""
    layout = dict(_weights.blob_layout(cfg))
    flat, seen = {}, {}
    for key, arr in tensors.items():
        name = _classify(key)
        if name is None or name not in layout:
            continue
        if name in flat and seen[name] != key:          # the same variable reached through two paths holds the same bytes
            if not np.array_equal(flat[name], np.asarray(arr, np.float32)):
                raise ValueError(f"{name}: keys {seen[name]!r} and {key!r} disagree")
            continue
        if tuple(arr.shape) != tuple(layout[name]):
            raise ValueError(f"{key}: shape {tuple(arr.shape)}, the configured model needs {tuple(layout[name])} for {name}")
        flat[name], seen[name] = np.asarray(arr, np.float32), key
    optional = {"W_q", "v_att"} if cfg.attention != "bahdanau" else set()
    missing = [n for n in layout if n not in flat and n not in optional]
    if missing:
        raise KeyError(f"checkpoint lacks {missing[:6]}{'...' if len(missing) > 6 else ''}; float tensors found: "
                       f"{sorted(k for k in tensors if k.endswith(_SUFFIX))[:8]} ...")
    for n in optional - set(flat):                       # Luong checkpoints carry no query layer: zeros keep the blob shape
        flat[n] = np.zeros(layout[n], np.float32)
    return flat


def flat_from_checkpoint(prefix: str, cfg: RvConfig) -> dict:
    return flat_from_tensors(read_tensor_bundle(prefix, want=lambda k: _classify(k) is not None), cfg)


def is_tf_checkpoint(path: str) -> bool:
    return os.path.exists(str(path) + ".index")


def main(argv=None):
    """python -m ravvent_basecaller_amd.checkpoint manifest [out.json] | convert <tf_prefix> <out.npz> [enc_depth dec_depth mode]"""
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "manifest":
        text = json.dumps({"config": {"enc_units": 128, "dec_units": 128, "enc_depth": 2, "dec_depth": 1, "vocab": 7},
                           "key_suffix": _SUFFIX, "segments": weights_manifest(RvConfig())}, indent=1)
        if len(argv) > 1:
            with open(argv[1], "w") as f:
                f.write(text + "\n")
        else:
            print(text)
        return 0
    if argv and argv[0] == "convert" and len(argv) >= 3:
        cfg = RvConfig(enc_depth=int(argv[3]) if len(argv) > 3 else 2, dec_depth=int(argv[4]) if len(argv) > 4 else 1,
                       mode=argv[5] if len(argv) > 5 else "joint")
        _weights.save(argv[2], cfg, flat_from_checkpoint(argv[1], cfg))
        return 0
    print(main.__doc__)
    return 2


if __name__ == "__main__":
    raise SystemExit(main())
