"""CPU tests of the host logic: weight blob layout, config mirror, tokenizer / chunk format,
evaluator slab splitting, shard ranges, and the C-ABI library's exported surface."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_weight_inventory_matches_survey(rv):
    cfg = rv.RvConfig()
    # 1,276,807 parameters for (128,128,enc 2,dec 1) + Bahdanau's W_q (16,384) + v (128)
    assert rv.weights.blob_size(cfg) == 1_276_807 + 16_384 + 128
    flat = rv.weights.init_weights(cfg, seed=1)
    blob = rv.weights.pack(cfg, flat)
    back = rv.weights.unpack(cfg, blob)
    assert all((back[k] == flat[k]).all() for k in flat)
    b = flat["enc_raw.0.fwd.b"]
    assert (b[128:256] == 1).all() and b[:128].sum() == 0 and b[256:].sum() == 0      # unit_forget_bias
    U = flat["enc_raw.0.fwd.U"].astype(np.float64)
    assert np.abs(U.T @ U - np.eye(512))[:128, :128].max() < 1.0                      # sane scale
    with pytest.raises(ValueError):
        rv.weights.unpack(cfg, blob[:-1])


def test_hash_init_is_exactly_reproducible(rv):
    cfg = rv.RvConfig()
    a = rv.weights.pack(cfg, rv.weights.init_weights(cfg, seed=7, scheme="hash"))
    b = rv.weights.pack(cfg, rv.weights.init_weights(cfg, seed=7, scheme="hash"))
    assert (a == b).all() and abs(float(a[:1000].astype(np.float64).sum()) - float(b[:1000].astype(np.float64).sum())) == 0
    u = rv.weights._splitmix_uniform(5, 4)
    assert np.allclose(u, rv.weights._splitmix_uniform(5, 4)) and (np.abs(u) <= 1).all()


def test_weight_file_roundtrip(rv, tmp_path):
    cfg = rv.RvConfig()
    flat = rv.weights.init_weights(cfg, seed=2)
    rv.weights.save(str(tmp_path / "w.npz"), cfg, flat)
    back = rv.weights.load(str(tmp_path / "w"), cfg)
    assert all((back[k] == flat[k]).all() for k in flat)


def test_config_struct_mirrors_header(rv):
    """Field order and count of the ctypes image == struct RvConfig in include/ravvent_hip.h."""
    hdr = open(os.path.join(ROOT, "include", "ravvent_hip.h")).read()
    body = hdr[hdr.index("typedef struct RvConfig {"):hdr.index("} RvConfig;")]
    names = re.findall(r"^\s*(?:int32_t|float)\s+(\w+);", body, re.M)
    assert names == [f[0] for f in rv.config.CRvConfig._fields_]
    assert ctypes.sizeof(rv.config.CRvConfig) == 4 * len(names)
    c = rv.RvConfig(mode="event", attention="bahdanau").to_c()
    assert (c.mode, c.attention, c.vocab, c.start_token, c.end_token) == (1, 1, 7, 2, 1)


def test_tokenizer_and_padding(rv):
    dl = rv.data_loader
    assert dl.nuc_tk.word_index == {"": 0, "^": 1, "$": 2, "a": 3, "c": 4, "g": 5, "t": 6}     # data_loader.py:21
    assert dl.nuc_tk.texts_to_sequences(["$ACGT^", "$xTT^"]) == [[2, 3, 4, 5, 6, 1], [2, 6, 6, 1]]
    assert dl.nuc_tk.sequences_to_texts([[2, 3, 0, 6, 1]]) == ["$ a  t ^"]
    p = dl.pad_input_snippets([np.ones((3, 5)), np.ones((40, 5))], 30)
    assert p.shape == (2, 30, 5) and p.dtype == np.float32 and p[0, 3:].sum() == 0 and p[1].sum() == 150
    t = dl.pad_sequences([[2, 3, 1], [2, 1]], dtype="int64", value=0)
    assert t.tolist() == [[2, 3, 1], [2, 1, 0]]
    assert (rv.utils.input_mask(p, 0.0)[0] == np.r_[np.ones(3, bool), np.zeros(27, bool)]).all()


def test_unpack_and_slab_split(rv):
    data = ("r", "e", "t")
    assert rv.utils.unpack_data_to_input_target(data, "raw") == ("r", "t")
    assert rv.utils.unpack_data_to_input_target(data, "event") == ("e", "t")
    assert rv.utils.unpack_data_to_input_target(data, "joint") == (("r", "e"), "t")
    parts = rv.evaluator.PerformanceEvaluator._split_into_chunks(np.arange(2500), 1024)
    assert [len(p) for p in parts] == [1024, 1024, 452]          # ravvent_performance_evaluator.py:19-22


def test_shard_ranges_cover_contiguously(rv):
    for n in (0, 1, 7, 256, 1025):
        for world in (1, 2, 3, 8):
            r = [rv.dist.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_library_exports_every_declared_symbol(rv):
    """The C-ABI library loads on a GPU-less host and exports exactly what the header declares;
    no compute entry point is called here."""
    hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("ravvent_hip.h", "ravvent_merge.h"))
    declared = set(re.findall(r"\b(rv_\w+)\s*\(", hdr)) - {"rv_handle"}
    lib = rv._capi.load_library()
    assert declared == {n for n, _, _ in rv._capi.SYMBOLS}
    for name in declared:
        assert hasattr(lib, name)
    assert lib.rv_abi_version() == 1


def test_no_cpu_fallback(rv):
    """Without a GPU the product path fails loudly (RV_EHIP), it never computes on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rv._capi.RavventHipError, match="no HIP device|RV_EHIP"):
        rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0)
    with pytest.raises(NotImplementedError):
        rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, rnn_type="gru")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ravvent-basecaller_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "ravvent_oracle" not in src, f


def test_mapping_evaluator_files_and_paf(rv, tmp_path):
    """ravvent_mapping_evaluator.py:74-108,130-167 without minimap2: FASTA/FASTQ writers, PAF identity, totals."""
    import json
    import ravvent_basecaller_amd.mapping_evaluator as me
    ev = me.MappingEvaluator(workdir=str(tmp_path))
    ev._create_fasta("ACGTACGTACGTAA", tmp_path / "r.fasta")
    ev._create_fastq("ACGTTT", tmp_path / "p.fastq")
    assert (tmp_path / "r.fasta").read_text() == ">ACGTACGTAC\nACGTACGTACGTAA"
    assert (tmp_path / "p.fastq").read_text() == "@ACGTTT\nACGTTT\n+\n!!!!!!"
    paf = tmp_path / "m.paf"
    paf.write_text("q\t900\t0\t400\t+\tt\t1000\t10\t420\t380\t410\t60\ttp:A:P\n"
                   "q\t900\t400\t900\t+\tt\t1000\t500\t990\t450\t500\t60\n"
                   "short\tline\n")
    r = ev._read_mapping_identity(paf)
    assert r == {"read_length": 900, "matches": 830, "total_block_len": 910, "identity": 830 / 910}
    (tmp_path / "empty.paf").write_text("")
    assert ev._read_mapping_identity(tmp_path / "empty.paf")["identity"] == 0.0
    res = [dict(r, ref_length=1000), {"read_length": 0, "matches": 0, "total_block_len": 0, "identity": 0.0, "ref_length": 500}]
    (tmp_path / "res.json").write_text(json.dumps(res))
    tot, valid, invalid = ev.compute_total_results(tmp_path / "res.json")
    assert (tot, valid, invalid) == (round(830 / 910 * 1000 / 1500 * 100, 3), round(830 / 910 * 100, 3), 50.0)
    import shutil
    if shutil.which("minimap2") is None:
        with pytest.raises(RuntimeError, match="minimap2"):
            ev.map_read("ACGT" * 10, "ACGT" * 9)


def test_tf_checkpoint_bundle_round_trip(rv, tmp_path):
    """checkpoint.py: a TF-format tensor bundle (LevelDB-format index + data shard) written with the reference model's
    variable paths reads back into exactly the weight blob it came from; optimizer slots / counters / the object graph
    are ignored; CRCs are checked; a wrong shape or a missing variable is an error that names it."""
    ck = rv.checkpoint
    cfg = rv.RvConfig(enc_depth=3, dec_depth=2, attention="bahdanau")
    flat = rv.weights.init_weights(cfg, seed=5)
    paths = ck.variable_paths(cfg)
    assert set(paths) == {n for n, _ in rv.weights.blob_layout(cfg)}
    tensors = {paths[n] + "/.ATTRIBUTES/VARIABLE_VALUE": a for n, a in flat.items()}
    tensors["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"] = np.array(1234, np.int64)
    tensors["decoder/fc/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE"] = np.zeros((128, 7), np.float32)
    tensors["save_counter/.ATTRIBUTES/VARIABLE_VALUE"] = np.array(3, np.int64)
    prefix = str(tmp_path / "model_chp")
    ck.write_tensor_bundle(prefix, tensors)
    assert ck.is_tf_checkpoint(prefix) and not ck.is_tf_checkpoint(str(tmp_path / "nothing"))
    back = ck.read_tensor_bundle(prefix)
    assert set(back) == set(tensors) and all(np.array_equal(back[k], tensors[k]) and back[k].dtype == tensors[k].dtype for k in tensors)
    got = ck.flat_from_checkpoint(prefix, cfg)
    assert np.array_equal(rv.weights.pack(cfg, got), rv.weights.pack(cfg, flat))
    # a Luong checkpoint has no query layer / attention_v: zeros keep the blob shape
    cfg_l = rv.RvConfig(enc_depth=3, dec_depth=2)
    luong = {k: v for k, v in tensors.items() if "query_layer" not in k and "attention_v" not in k}
    got_l = ck.flat_from_tensors(luong, cfg_l)
    assert not got_l["W_q"].any() and np.array_equal(got_l["W_mem"], flat["W_mem"])
    # manifest: offsets tile the blob in blob order
    man = ck.weights_manifest(cfg)
    assert [m["segment"] for m in man] == [n for n, _ in rv.weights.blob_layout(cfg)]
    assert man[-1]["offset_floats"] + man[-1]["count"] == rv.weights.blob_size(cfg)
    blob = rv.weights.pack(cfg, flat)
    m = man[7]
    assert np.array_equal(blob[m["offset_floats"]:m["offset_floats"] + m["count"]].reshape(m["shape"]), flat[m["segment"]])
    # error paths
    bad = dict(tensors); bad[paths["W_fc"] + "/.ATTRIBUTES/VARIABLE_VALUE"] = np.zeros((128, 5), np.float32)
    with pytest.raises(ValueError, match="shape"):
        ck.flat_from_tensors(bad, cfg)
    few = {k: v for k, v in tensors.items() if "encoder_event/rnn_layers/1/backward_layer/cell/bias" not in k}
    with pytest.raises(KeyError, match="enc_event.1.bwd.b"):
        ck.flat_from_tensors(few, cfg)
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); raw[100] ^= 0xFF
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="CRC32C"):
        ck.read_tensor_bundle(prefix)
    # the committed manifest is the one the code generates for the default model
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "weights_manifest.json")) as f:
        assert json.load(f)["segments"] == ck.weights_manifest(rv.RvConfig())


def test_bench_self_launch_command(monkeypatch):
    """bench.py --gpus N without a launcher starts torch.distributed.run as a CHILD process (never an exec) with the
    same arguments, on 127.0.0.1, relays its stdout and returns its exit code."""
    import importlib.util
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7, stdout='[Gloo] Rank 0 is connected\n{"metric": "x"}\n')
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import io
    out = io.StringIO()
    monkeypatch.setattr(bench.sys, "stdout", out)
    rc = bench.self_launch(types.SimpleNamespace(gpus=4), ["--gpus", "4", "--steps", "3", "--strong"])
    monkeypatch.undo()
    assert rc == 7 and out.getvalue() == '{"metric": "x"}\n'      # exactly one JSON line reaches stdout
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == ["--gpus", "4", "--steps", "3", "--strong"] and cmd[-6].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # executed-FLOP accounting of the decode: chunk_steps replace B * S
    assert bench.algorithmic_flops("dec_persist", 4, 300, 30, 5, 47) == 4 * 5 * 47 * (369408 + 768 * 330)
    assert bench.algorithmic_flops("dec_persist", 4, 300, 30, 5, 47, [47, 10, 12, 47]) == 5 * 116 * (369408 + 768 * 330)


def test_split_f16_operands_hold_fp32_products():
    """The arithmetic behind `split_projection` / `matrix_attention` ("split operands"), restated in numpy: a
    value scaled below 2^14 and cut as v = f16(v) + f16(v - f16(v)) is held to 2^-23 of itself, every f16 x f16 part product is exact
    in f32, and x.w = xh.wh + xh.wl + xl.wh is no further from the fp64 product than an f32 FMA chain over the same operands."""
    rng = np.random.default_rng(5)
    M, K, N = 256, 256, 64
    x = (np.tanh(rng.normal(0, 0.7, (M, K))) * rng.uniform(0, 1, (M, K))).astype(np.float32)       # |x| <= 1, like an LSTM output
    w = rng.uniform(-0.09, 0.09, (K, N)).astype(np.float32)
    w[:, :4] *= 30.0; w[:, 4:8] *= 1e-3; w[7, 20] = 4.0                                              # large / tiny columns, an outlier
    sx = np.float32(2.0 ** 14)
    sw = (2.0 ** (14 - np.ceil(np.log2(np.abs(w).max(axis=0) * 1.0001)))).astype(np.float32)        # power of two per column

    def split(v):
        hi = v.astype(np.float16)
        lo = (v - hi.astype(np.float32)).astype(np.float16)
        return hi, lo

    xs, ws = x * sx, w * sw
    assert np.abs(xs).max() < 65504 and np.abs(ws).max() < 65504 and np.abs(ws).max(axis=0).min() >= 2.0 ** 13 / 1.0001
    xh, xl = split(xs); wh, wl = split(ws)
    for v, h, l in ((xs, xh, xl), (ws, wh, wl)):
        err = np.abs(v.astype(np.float64) - h.astype(np.float64) - l.astype(np.float64))
        big = np.abs(v) >= 2.0 ** -3                      # (below that the low part reaches the f16 subnormals: 2^-25 absolute, 2^-39 of the 2^14 range)
        assert (err[big] <= np.abs(v[big]) * 2.0 ** -22).all() and (err[~big] <= 2.0 ** -25).all()
    # part products are exact in f32: 11-bit x 11-bit significands
    p32 = xh[:, :1].astype(np.float32) * wh[:1, :].astype(np.float32)
    assert (p32.astype(np.float64) == xh[:, :1].astype(np.float64) * wh[:1, :].astype(np.float64)).all()
    d = lambda a, b: a.astype(np.float64) @ b.astype(np.float64)
    got = ((d(xh, wh) + d(xh, wl) + d(xl, wh)) / (np.float64(sx) * sw.astype(np.float64))).astype(np.float32)
    ref = d(x, w)
    chain = np.zeros((M, N), np.float32)
    for k in range(K):
        chain = chain + x[:, k:k + 1] * w[k:k + 1, :]
    e_split, e_chain = np.abs(got - ref), np.abs(chain - ref)
    assert e_split.max() <= e_chain.max() and e_split.mean() <= e_chain.mean()
