#!/usr/bin/env python3
"""bench.py -- throughput of the Ravvent hot path (Basecaller.beam_search_prediction) on MI355X.

  python bench.py --gpus N --steps K --warmup W [--strong]

With N > 1 and no launcher in the environment this script starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...` as a CHILD process (before
anything here touches the GPU), relays its JSON line and exits with its code; started under a launcher
(WORLD_SIZE set) it is one rank of that job.

A "step" is one pass of the hot path over one batch of synthetic chunks: workload C3 of BASELINE.json (joint
raw+event mode, 300-sample raw windows + 30 events, beam 5, max_output_len 48), inputs resident in HBM before the
timed region, outputs left in HBM.  The K timed steps go through the library's asynchronous calls (rv_beam_search_submit_dev /
rv_beam_search_collect_dev; `Basecaller.beam_search_stream`): --depth of them are in flight at a time, every step is collected
inside the timed region, results are byte-identical to the synchronous call (`--depth 0` times that one instead, and its rate
is reported beside the headline as `synchronous`).
  weak scaling (default): 256 chunks per GPU per step; with N > 1 the global slab of N x 256 chunks goes through the
    shipped multi-GPU path `dist.sharded_beam_search` (contiguous chunk shards, ONE RCCL all-gather per step).
  --strong: a fixed read of --read-chunks (8,192) chunks per step, sharded N ways, every rank decoding its shard
    in slabs of 256 -- what BASELINE configs 4/5 describe; same single gather.
Rank 0 prints ONE JSON line.

metric: kbases/s = chunks/s * 6 bases per chunk / 1000 (NOMINAL: stride 6 events => ~6 new bases per chunk,
/root/reference/ravvent_performance_evaluator.py:16); chunks/s is reported beside it, and
`read_level` holds the reference's own figure -- bases_num (reference length) / total_processing of the evaluator call
sequence (:79,86,125) -- on one synthetic long read with base-emitting weights.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BASES_PER_CHUNK = 6
PEAK_F32_TFLOPS = 157.3     # MI355X fp32 vector = fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_NOTE = ("fp32 matrix / vector peak, the arithmetic type the path's results carry.  Every dense contraction of the path -- encoder "
             "recurrences and input projections, memory projection, Luong scores / context and the decoder's cell product -- takes each fp32 "
             "product as three exact f16 part products on v_mfma_f32_16x16x32_f16 (the 2.5 PFLOP/s pipe); gate math, softmax, output layer "
             "and beam step run on the fp32 VALU.  FLOPs counted are the algorithm's fp32 FLOPs, not MFMA operations, so a fraction "
             "of this peak near or above 1 says how much of the work left the fp32 lanes, not that a limit was reached")
PEAK_F16_MFMA_TFLOPS = 2500.0   # dense f16 / bf16 MFMA peak (MI355X_MICROARCH.md): what the split-operand GEMMs issue their part products on
PEAK_HBM_GBS = 8000.0       # HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s achievable)
DEC_FLOPS = lambda Tm: 369408 + 768 * Tm     # per beam row per decode step


def algorithmic_flops(kernel, B, T_r, T_e, W, S, chunk_steps=None, wide=False):
    """Algorithmic FLOPs of ONE launch of `kernel` (2 per MAC, pointwise ignored).  The decode is credited
    B*W*S row-steps like the reference's slab-wide loop; `chunk_steps` (steps every chunk really ran in the persistent
    decode, which leaves a chunk once its beams are finished) gives the executed count instead."""
    Tm = T_r + T_e
    rec = 2 * 128 * 512 * 2            # recurrent product, both directions, per chunk-step
    rows = B * S if chunk_steps is None else int(sum(chunk_steps))
    proj = 0 if wide else 256 * 1024 * 2    # matrix-pipe recurrence: the layer >= 1 input projection is its own launch (gemm_inproj_*)
    return {
        "lstm_rec_raw_l0": B * T_r * (rec + 2 * 1 * 512 * 2),
        "lstm_rec_event_l0": B * T_e * (rec + 2 * 5 * 512 * 2),
        "lstm_rec_raw_l1p": B * T_r * (rec + proj),      # recurrence (+ the fused input projection of the packed-FMA form)
        "lstm_rec_event_l1p": B * T_e * (rec + proj),
        "gemm_inproj_raw": B * T_r * 256 * 1024 * 2,     # both directions per launch
        "gemm_inproj_event": B * T_e * 256 * 1024 * 2,
        "gemm_keys": B * Tm * 256 * 128 * 2,
        "gemm_memory": B * Tm * 256 * 256 * 2,               # [keys | attention-layer image of the values] for the persistent decode
        "decode_graph": W * rows * DEC_FLOPS(Tm),
        "dec_persist": W * rows * DEC_FLOPS(Tm),              # the whole decode loop is one launch
        "dec_cell": B * W * 2 * 256 * 512,
    }.get(kernel)


def algorithmic_bytes(kernel, B, T_r, T_e, W, S):
    """Algorithmic HBM bytes of ONE launch of an HBM-bound kernel."""
    Tm = T_r + T_e
    return {
        # single-pass Luong attend: the chunk's values [T_m,256] fp32 once per step (+ mask bytes)
        "dec_attend": B * Tm * (256 * 4 + 1),
    }.get(kernel)


def cpu_baseline(rv, cfg, flat, T_r, T_e, W, L, sample_chunks=0, target_s=12.0, eager_s=6.0):
    """The C restatement of the oracle (oracle/ravvent_cpu.c, 'port') timed on this host's cores on a bounded sample of
    the same workload: a 256-chunk probe sizes the sample to ~target_s seconds of CPU work (sample_chunks > 0 fixes it
    instead).  Beside it, as a sanity band, a torch-CPU eager engine at TF-eager op granularity."""
    from oracle import cpu_port, torch_eager          # checker / baseline only
    import torch
    blob = rv.weights.pack(cfg, flat)
    cores = cpu_port.max_threads()
    run = lambda r, e: cpu_port.run(cfg.oracle_cfg(), cfg.enc_depth, cfg.vocab, blob, r, e, W, L)
    raw, ev, _ = rv.synthetic.make_slab(256, T_r, T_e, seed=100)
    run(raw[:32], ev[:32])                            # warm-up (thread pool, page faults)
    t0 = time.perf_counter(); run(raw, ev); probe = time.perf_counter() - t0
    n = sample_chunks or int(min(max(256, round(target_s / probe) * 256), 16384))
    raw, ev, _ = rv.synthetic.make_slab(n, T_r, T_e, seed=101)
    t0 = time.perf_counter()
    tok, _ = run(raw, ev)
    dt = time.perf_counter() - t0
    out = {"value": round(n / dt * BASES_PER_CHUNK / 1000.0, 4), "unit": "kbases/s",
           "chunks_per_s": round(n / dt, 2), "cores": cores, "kind": "port",
           "sample": f"{n} chunks of the same workload (joint {T_r}+{T_e}, beam {W}, L {L}) in slabs of the "
                     f"C port's choosing, {dt:.1f} s wall on {cores} threads, S={tok.shape[1]}"}
    try:   # torch eager band: slabs of 256 like the GPU run, torch's own intra-op threads
        w = rv.weights.flat_to_nested(cfg, flat)
        raw, ev, _ = rv.synthetic.make_slab(256, T_r, T_e, seed=102)
        torch_eager.beam_search(w, cfg.oracle_cfg(), raw[:8], ev[:8], W, 4)     # warm-up
        t0 = time.perf_counter(); k = 0
        while True:
            torch_eager.beam_search(w, cfg.oracle_cfg(), raw, ev, W, L); k += 1
            if time.perf_counter() - t0 > eager_s or k >= 8:
                break
        de = time.perf_counter() - t0
        out["torch_eager_band"] = {"value": round(256 * k / de * BASES_PER_CHUNK / 1000.0, 4), "unit": "kbases/s",
                                   "chunks_per_s": round(256 * k / de, 2), "cores": torch.get_num_threads(),
                                   "sample": f"{k} slab(s) of 256 chunks, {de:.1f} s wall, torch {torch.__version__} CPU eager, "
                                             f"one op per TF-eager op (oracle/torch_eager.py)"}
    except Exception as e:  # the band is a sanity figure: never lose the bench line over it
        out["torch_eager_band"] = {"error": repr(e)}
    return out


def read_level(rv, device, n_bases=49500, slab=1024, pipelined=False, concurrent=1):
    """The reference's read-level metric (ravvent_performance_evaluator.py:79,86,125): bases_num / (t_predicting +
    t_postprocessing + t_merge) with bases_num = the read's REFERENCE length, through the evaluator call sequence on one
    synthetic long read (`synthetic.make_read`: ~8,200 chunks of <= 200 samples + <= 30 events, stride 6, cut by the
    chunker), slabs of 1024 like the reference evaluator, fused on-device post-processing + C++ merger; weights that emit
    full-length varied base strings (`weights.base_calling_weights`; untrained weights cannot call the read correctly:
    `merged_bases` / `merged_over_reference` say how much the merger stitched, they are not an accuracy).  Host buffers
    in, merged read out; chunking / event detection are outside the metric as in the reference (:32-45)."""
    import numpy as np
    T_r, T_e = 200, 30
    sig, lab = rv.synthetic.make_read(n_bases, seed=0)
    la = np.asarray(lab)
    raw, ev, nuc = rv.data_loader.snippets_to_slab(*rv.data_loader.prepare_snippets(sig, la[:, :2].astype(int), la[:, 2], 6))
    L = int(nuc.shape[1])
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=slab, max_raw_len=T_r, max_event_len=T_e,
                       max_output_len=max(L, 2), device=device)
    # calls of L - 1 varied letters: every chunk is longer than the merger's 25-letter overlap, so the read really is spliced and grows
    # (the round-2 weights emitted calls of <= 25 letters: 7.6 k alignments ran and the merged read stayed at 145 bases)
    bc.set_weights_flat(rv.weights.base_calling_weights(bc.cfg))
    e = rv.evaluator.PerformanceEvaluator(bc, pipelined_merge=pipelined, concurrent_slabs=concurrent)   # (fused post-processing: the default)
    e.run_slabs(raw[:slab], ev[:slab], nuc[:slab], chunk_size=slab)          # warm-up
    best = None
    for _ in range(3):
        r = e.run_slabs(raw, ev, nuc, bases_num=n_bases, chunk_size=slab)
        if best is None or r["total_processing"] < best["total_processing"]:
            best = r
    e.close()
    bc.close()
    tp, n_chunks = best["total_processing"], best["chunks_num"]
    return {"workload": f"one synthetic read of {n_bases} bases -> {n_chunks} chunks (joint <=200+<=30, stride 6), beam 5, L {L}, slabs "
                        f"of {slab}, base-emitting weights, fused post-processing + C++ merger"
                        + (" pipelined behind the GPU" if pipelined else "")
                        + f", {max(concurrent, 2)} slabs in flight (asynchronous calls)",
            "kbases_per_s": round(n_bases / tp / 1000.0, 2), "chunks_per_s": round(n_chunks / tp, 1),
            "bases_num": n_bases, "merged_bases": len(best["merged_seq"]),
            "merged_over_reference": round(len(best["merged_seq"]) / n_bases, 3),
            "call_letters_min": int(min(len(s) for s, _ in best["nuc_preds"])),
            "t_predicting": round(best["t_predicting"], 5), "t_postprocessing": round(best["t_postprocessing"], 5),
            "t_merge": round(best["t_merge"], 5), "total_processing": round(tp, 5)}


def variant_timing(rv, device, B, T_r, T_e, W, L, what, steps=10, **ctor):
    """The same workload on another model variant of the reference (untimed extra, own handle): `bahdanau` -- north_star names Bahdanau
    attention (Decoder(attention_type='bahdanau'), basecaller.py:131-132): tanh scores on the vector ALU, everything else on the matrix
    pipe; `enc3_dec2` -- the reference's best model family (three BiLSTM layers per encoder, two stacked decoder cells,
    basecaller.py:85-91; accuracy_results_all.lambda.beam5.json).  Synchronous slabs with the decode launch timed alone, then streamed
    through the asynchronous calls like the headline (10 slabs in flight; results identical)."""
    import gc
    import torch
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, honor_attention_type=True,
                       max_batch=B, max_raw_len=T_r, max_event_len=T_e, max_output_len=L, device=device, **ctor)
    bc.init_random_weights(seed=22)
    raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=0)
    x = (torch.from_numpy(raw).to(bc.device), torch.from_numpy(ev).to(bc.device))
    bc.reuse_output_buffers = True
    bc.set_option("profile", 3)
    gc.collect(); gc.disable()
    for _ in range(3):
        tok, _ = bc.beam_search_prediction(x, W, L)
    bc.reset_profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tok, _ = bc.beam_search_prediction(x, W, L)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = bc.profile()
    bc.set_option("profile", 0)
    bc.set_async_depth(10)
    for _ in bc.beam_search_stream((x for _ in range(10)), W, L):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in bc.beam_search_stream((x for _ in range(4 * steps)), W, L):
        pass
    torch.cuda.synchronize()
    dts = (time.perf_counter() - t0) / (4 * steps)
    gc.enable()
    bc.close()
    name = "dec_persist" if "dec_persist" in prof else "decode_graph"
    return {"workload": f"C3 shape, {what}, " + ("one-launch persistent decode" if name == "dec_persist" else "per-step decode kernels in a hipGraph"),
            "ms_per_step": round(dt * 1e3, 4), "chunks_per_s": round(B / dt, 1), "decode_steps": int(tok.shape[1]),
            "decode_ms_per_launch": round(prof[name][0] / max(prof[name][1], 1), 4),
            "streamed": {"ms_per_step": round(dts * 1e3, 4), "chunks_per_s": round(B / dts, 1), "note": f"{4 * steps} slabs through the asynchronous calls, 10 in flight"}}


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: run the job as a child `torch.distributed.run` (never an exec: this process may
    not replace itself once anything has touched the GPU, and a child keeps that rule trivially true)."""
    port = int(os.environ.get("MASTER_PORT", 29500 + os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout.splitlines():          # ONE JSON line on stdout; whatever else the ranks printed there (gloo's connection
        if line.startswith('{"metric"'):        # notices, ...) goes to stderr
            sys.stdout.write(line + "\n")
        else:
            sys.stderr.write(line + "\n")
    sys.stdout.flush()
    return p.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="chunks per GPU per slab (C3: 256)")
    ap.add_argument("--raw-len", type=int, default=300)
    ap.add_argument("--event-len", type=int, default=30)
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--max-output-len", type=int, default=48)
    ap.add_argument("--recurrence", default="mx", choices=["mx", "fma", "auto"],
                    help="library option wide_recurrence: mx = matrix-pipe recurrence, 16 chunks per workgroup (library default); "
                         "fma = packed-FMA kernels; auto = per call by chunks in flight")
    ap.add_argument("--depth", type=int, default=10, help="slabs in flight through the asynchronous calls (1..16); 0 = the synchronous call")
    ap.add_argument("--gather-per-step", action="store_true", help="N > 1: one all-gather per step (dist.sharded_beam_search_stream) instead of "
                    "ONE at the end of the steps (dist.sharded_beam_search_many, the default)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: with --gpus 1, run the N > 1 code path in a one-rank process group")
    ap.add_argument("--strong", action="store_true", help="strong scaling: a fixed read of --read-chunks chunks per step, sharded over the GPUs")
    ap.add_argument("--read-chunks", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the read_level and bahdanau sub-lines (N=1 only)")
    ap.add_argument("--per-step-decode", action="store_true",
                    help="A/B: per-step decode kernels in a hipGraph instead of the one-launch persistent decode")
    ap.add_argument("--attend-threads", type=int, default=0, help="0 auto | 256 | 512 (library option attend_threads)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL; the real path) | gloo (control-flow rehearsal: ranks may share a GPU, the gather goes through host memory)")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the per-kernel event pass over the decode loop")
    ap.add_argument("--cpu-sample", type=int, default=0, help="chunks for the CPU baseline (0 = size to ~12 s)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist
    import ravvent_basecaller_amd as rv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dist_backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist_path = world > 1 or args.force_dist      # --force-dist: the N > 1 code path (process group, shard -> decode -> gather) with ONE rank
    if dist_path:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    B, T_r, T_e, W, L = args.batch, args.raw_len, args.event_len, args.beam, args.max_output_len
    bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", rv.data_loader.INPUT_PADDING,
                       encoder_depth=2, decoder_depth=1, rnn_type="bilstm", attention_type="luong",
                       beam_width=W, device=local, max_batch=B, max_raw_len=T_r, max_event_len=T_e,
                       max_output_len=L)
    flat = bc.init_random_weights(seed=22)            # Keras-default initialisers, seed as ravvent.py:9
    bc.reuse_output_buffers = True                    # explicit opt-in: no allocator traffic in the timed loop
    if args.attend_threads:
        bc.set_option("attend_threads", args.attend_threads)
    if args.per_step_decode:
        bc.set_option("persistent_decode", 0)
    # the global batch of a step: every rank holds all of it (inputs are ~2 KB per chunk) and decodes its contiguous shard
    n_global = args.read_chunks if args.strong else world * B
    raw, ev, _ = rv.synthetic.make_slab(n_global, T_r, T_e, seed=0)
    d_raw, d_ev = torch.from_numpy(raw).to(dev), torch.from_numpy(ev).to(dev)
    lo, hi = rv.dist.shard_range(n_global, rank, world)

    depth = max(0, min(args.depth, 16))
    if depth:
        bc.set_async_depth(depth)
    bc.set_option("wide_recurrence", {"mx": 1, "fma": 0, "auto": -1}[args.recurrence])
    plain_steps = not dist_path or bool(os.environ.get("RV_BENCH_PLAIN_STEPS"))     # (diagnostic switch: process group up, steps without it)
    cut = lambda g: [(g[0][a:a + B], g[1][a:a + B]) for a in range(lo, hi, B)]      # this rank's slabs of one global batch
    # the warm-up runs on DISTINCT global batches (other seeds), so that every slab context's buffers hold something else when the timed
    # region starts: a result read from the wrong context or a stale buffer cannot pass the check after the timed region
    warm_batches = []
    for i in range(3):
        rw, ew, _ = rv.synthetic.make_slab(n_global, T_r, T_e, seed=7001 + i)
        warm_batches.append((torch.from_numpy(rw).to(dev), torch.from_numpy(ew).to(dev)))

    def run_steps(k, batches=None, gather_per_step=args.gather_per_step):
        """k steps; returns (the results of every step, per-step completion times).  depth > 0: the steps' slabs stream through the
        asynchronous calls, `depth` in flight; a step is complete when its last slab is collected.  N > 1: the shipped multi-GPU path,
        shard -> decode -> ONE all-gather (RCCL over xGMI) at the end of the steps (or one per step, the next step's shard submitted
        before this one is gathered).  `batches`: the global batches the steps cycle through (default: the bench batch, every step)."""
        batches = batches or [(d_raw, d_ev)]
        glob = [batches[i % len(batches)] for i in range(k)]
        outs, stamps = [], []
        if not plain_steps:
            if depth and not gather_per_step:     # every rank streams its shards of all k steps, ONE all-gather at the end
                outs = list(rv.dist.sharded_beam_search_many(bc, glob, W, L, slab=B, reuse_buffers=True))
                stamps = [time.perf_counter()] * k
            elif depth:
                for out in rv.dist.sharded_beam_search_stream(bc, iter(glob), W, L, slab=B):
                    outs.append(out); stamps.append(time.perf_counter())
            else:
                for g in glob:
                    outs.append(rv.dist.sharded_beam_search(bc, g[0], g[1], W, L, slab=B))
                    stamps.append(time.perf_counter())
            return outs, stamps
        if depth:
            n, per = 0, max(len(cut(glob[0])), 1)
            for out in bc.beam_search_stream((x for g in glob for x in cut(g)), W, L):
                n += 1
                if n % per == 0:
                    outs.append(out); stamps.append(time.perf_counter())
            return outs, stamps
        for g in glob:
            for x in cut(g):
                out = bc.beam_search_prediction(x, beam_width=W, max_output_len=L)
            outs.append(out); stamps.append(time.perf_counter())
        return outs, stamps

    def fence():
        if dist_path:
            dist.barrier()
        torch.cuda.synchronize()

    # No profiling inside the timed region: two hipEvents per slab on the decode launch (option profile = 3, what rounds 2-3 timed with)
    # cost the stream ~5 % (341 k against 358-373 k chunks/s at these settings, tools/graph_ab.py).  The dominant launch is timed alone
    # (`roofline`) and inside an untimed copy of the stream (`span_ms_in_stream`) right after the timed region.
    bc.set_option("profile", 0)
    import gc
    gc.collect(); gc.disable()             # before the warm-up: a collection between warm-up and timing idles the GPU (~50 ms)
    if dist_path and depth and not args.gather_per_step:   # the gather buffers of the K timed steps, sized ahead of time (a first-use
        rv.dist.reserve_many_buffers(bc, args.steps, n_global, L)   # allocation inside the timed region costs more than the collective)
    if dist_path:                          # the FIRST use of a collective sets it up (tens of ms with RCCL): not between the warm-up and the
        dist.barrier()                     # timed region, where the idle GPU would drop its clocks (RV_BENCH_GAP_MS shows what a gap there costs)
        _g = torch.zeros((world, 8), dtype=torch.int32, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_gather_into_tensor(_g, _g[rank:rank + 1].clone())
        del _g
    run_steps(max(args.warmup, 1), warm_batches)   # (>=1: contexts, graph capture and event pools are built here, not in the timed region)
    for _ in range(int(os.environ.get("RV_BENCH_PRERUN", "0"))):      # diagnostic: whole untimed passes of the K steps first
        run_steps(args.steps)
    if plain_steps and depth:              # the K result pairs the timed region keeps (for the check after it): their allocator blocks exist already
        _pre = [torch.empty((2, B, max(L - 1, 0)), dtype=torch.int32, device=dev) for _ in range(args.steps * max(len(cut((d_raw, d_ev))), 1))]
        del _pre
    bc.reset_profile()
    fence()
    if os.environ.get("RV_BENCH_GAP_MS"):          # diagnostic: an idle gap between the warm-up and the timed region
        time.sleep(float(os.environ["RV_BENCH_GAP_MS"]) * 1e-3)
    t0 = time.perf_counter()
    timed_outs, stamps = run_steps(args.steps)
    t_run = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    # `reuse_output_buffers` / the reused gather buffers: results are views that the next call with the same shapes overwrites -> copy now
    timed_outs = [(t.clone(), s_.clone()) for t, s_ in timed_outs]
    if os.environ.get("RV_BENCH_VERBOSE"):
        print(f"timed region: run_steps {t_run * 1e3:.3f} ms, closing fence {(dt - t_run) * 1e3:.3f} ms", file=sys.stderr)
    gc.enable()
    per_step = [b - a for a, b in zip([t0] + stamps[:-1], stamps)]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the decode launch inside the stream (hipEvents on the library's stream around it, all slabs in flight around them): an untimed copy of
    # the timed region with option profile = 3
    prof_dec = {}
    if plain_steps and depth:
        bc.set_option("profile", 3)
        run_steps(depth)
        bc.reset_profile()
        torch.cuda.synchronize()
        run_steps(args.steps)
        torch.cuda.synchronize()
        prof_dec = bc.profile()
        bc.set_option("profile", 0)
    # Untimed extra, not `value`: the same stream of steps once the pipeline and the chip's clocks have settled.  The timed region above
    # starts from an idle GPU (the contract's synchronisation) and holds the pipeline's fill and drain; a service streams for minutes.
    steady = None
    if not dist_path and depth and not args.no_extras:
        n_ss = 4 * max(args.steps, 10)
        gc.collect(); gc.disable()
        run_steps(max(args.steps, 10))
        torch.cuda.synchronize()
        ts = time.perf_counter()
        run_steps(n_ss)
        torch.cuda.synchronize()
        steady = (time.perf_counter() - ts) / n_ss
        gc.enable()
    gather_step_ms = None
    if not plain_steps and depth and not args.gather_per_step and not args.no_extras:     # every rank takes part: collectives inside
        run_steps(max(args.warmup, 1), None, True)
        fence()
        ts = time.perf_counter()
        run_steps(args.steps, None, True)
        fence()
        gather_step_ms = (time.perf_counter() - ts) / args.steps
    slabs_per_step = -(-(hi - lo) // B)            # this rank's launches of each kernel per step
    x0 = (d_raw[lo:lo + B], d_ev[lo:lo + B])       # this rank's first slab: what the untimed per-kernel passes run on
    Bk = int(x0[0].shape[0])
    chunk_steps, prof_iso, sync_ms, sync_auto_ms, verified = None, {}, None, None, None
    # which recurrence form the timed region ran (auto: the library's per-call rule on the chunks in flight)
    n_fl = Bk * max(depth, 1)
    form_used = {"mx": 1, "fma": 0, "auto": 0 if n_fl < 160 else (2 if n_fl <= 512 else 1)}[args.recurrence]     # (the library's rule for -1)
    wide_used = form_used != 0
    if rank == 0:
        # every launch ALONE on the chip, in a short untimed pass of synchronous calls with the timed region's kernel selection
        # (local decode only: no collective here)
        bc.set_option("wide_recurrence", form_used)
        bc.set_option("profile", 1)
        bc.reset_profile()
        for _ in range(10):
            tk0, _ = bc.beam_search_prediction(x0, beam_width=W, max_output_len=L)
        prof_iso = {k: v[0] / max(v[1], 1) for k, v in bc.profile().items()}        # ms per launch
        S = int(tk0.shape[1])
        if not args.per_step_decode:
            chunk_steps = bc.get_tensor("chunk_steps").astype(int)
        bc.set_option("profile", 0)
        # ---- the check of what the timed region returned (untimed): EVERY timed step's tokens and scores against a synchronous call on
        # the same input -- byte for byte when the recurrence form is fixed (mx / fma), to f32 rounding with `auto` (the synchronous call
        # then picks the other form).  N > 1: the gathered global batch against this rank decoding all of it, slab by slab.
        want_t, want_s = [], []
        for a in range(0, n_global, B):
            t_, s_ = bc.beam_search_prediction((d_raw[a:a + B], d_ev[a:a + B]), beam_width=W, max_output_len=L)
            want_t.append(t_.clone()); want_s.append(s_.clone())
        S_all = max(int(t_.shape[1]) for t_ in want_t)
        end_tok = int(bc.output_end_token)
        def widen(t_, s_):      # a slab that stopped before the global batch's longest one: what the slab-wide loop would have emitted
            if t_.shape[1] == S_all or t_.shape[1] == 0:
                return t_, s_
            pad = S_all - t_.shape[1]
            return (torch.cat([t_, torch.full((t_.shape[0], pad), end_tok, dtype=t_.dtype, device=t_.device)], 1),
                    torch.cat([s_, s_[:, -1:].expand(-1, pad)], 1))
        if plain_steps:
            want = (want_t[-1], want_s[-1]) if len(want_t) == 1 else None
        else:
            ws = [widen(t_, s_) for t_, s_ in zip(want_t, want_s)]
            want = (torch.cat([w_[0] for w_ in ws], 0), torch.cat([w_[1] for w_ in ws], 0))
        n_ok, exact = 0, args.recurrence != "auto"
        for t_, s_ in timed_outs:
            w_t, w_s = want if want is not None else (want_t[-1], want_s[-1])       # (plain steps keep the step's LAST slab)
            w_t, w_s = w_t.to(t_.device), w_s.to(s_.device)
            if tuple(t_.shape) != tuple(w_t.shape):
                continue
            if exact:
                n_ok += int(torch.equal(t_, w_t) and torch.equal(s_.view(torch.int32), w_s.view(torch.int32)))
            else:
                rows = (t_ == w_t).all(dim=1)
                n_ok += int(rows.float().mean().item() >= 0.98 and (not rows.any() or (s_[rows] - w_s[rows]).abs().max().item() < 1e-4))
        verified = {"ok": n_ok == len(timed_outs) and len(timed_outs) == args.steps, "steps_checked": len(timed_outs), "steps_equal": n_ok,
                    "how": ("tokens and score bits of every timed step == a synchronous call on the same input" if exact else
                            "every timed step vs a synchronous call (other recurrence form): >= 98 % rows token-identical, their scores within 1e-4")
                           + "; the warm-up ran on other inputs, so every context held different data when the timed region started"}
        if world == 1:                 # the synchronous call's rate, beside the headline (untimed extras): the headline's recurrence form,
            def time_sync(n=20):       # and the per-call choice of form (option wide_recurrence = -1)
                gc.collect(); gc.disable()
                for _ in range(3):
                    bc.beam_search_prediction(x0, beam_width=W, max_output_len=L)
                torch.cuda.synchronize()
                ts = time.perf_counter()
                for _ in range(n):
                    bc.beam_search_prediction(x0, beam_width=W, max_output_len=L)
                torch.cuda.synchronize()
                gc.enable()
                return (time.perf_counter() - ts) / n * 1e3
            sync_ms = time_sync()
            bc.set_option("wide_recurrence", -1)
            sync_auto_ms = time_sync()
    if os.environ.get("RV_BENCH_VERBOSE"):
        print("per-step ms:", " ".join(f"{x*1e3:.2f}" for x in per_step), file=sys.stderr)

    if rank == 0:
        chunks_per_s = n_global * args.steps / dt
        # Per-kernel view of the decode graph: a short extra pass with hipEvents around EVERY kernel
        # (option profile=2: the graph is bypassed, same kernels, same stream).  Not part of `value`.
        dec = {}
        if not args.no_kernel_pass and args.per_step_decode:
            bc.set_option("profile", 2)
            bc.reset_profile()
            for _ in range(3):     # local decode only: no collective here (this block runs on rank 0 alone)
                bc.beam_search_prediction(x0, beam_width=W, max_output_len=L)
            dec = {k: v for k, v in bc.profile().items() if k.startswith("dec_") and k != "dec_finalize"}
            bc.set_option("profile", 1)
        per_slab = dict(prof_iso)                  # ms per launch of every kernel name, alone on the chip
        if dec:
            per_slab.pop("decode_graph", None)
            for k, (ms, n) in dec.items():
                per_slab[k] = ms / 3.0
        # workgroups per launch -> the share of the chip a launch can use, and from it the kernel's CU-time per slab: with several
        # slabs in flight that, not the launch duration, is what a kernel costs the whole job
        Tm = T_r + T_e
        def ws_wgs(M):                  # launch_gemm_split_blocks' grid for the weight-stationary form (8 half column blocks, 12 waves)
            k, ntile = 4, -(-M // 32)
            while k > 1 and 8 * (k - 1) * 12 >= ntile:
                k -= 1
            return 8 * k * 8
        wgs = {"dec_persist": Bk, "dec_finalize": 0, "input_mask": 0,
               "gemm_memory": min(256, -(-(Bk * Tm) // 128)),
               "gemm_inproj_raw": ws_wgs(Bk * T_r) if wide_used else min(256, -(-(Bk * T_r) // 128)),
               "gemm_inproj_event": ws_wgs(Bk * T_e) if wide_used else min(256, -(-(Bk * T_e) // 128)), "inproj_event_l0": 256}
        def rows_per_block(b):          # the library's pick_rows_per_block
            bt, r = (2 * b + 255) // 256, 1
            while r < bt and r < 8:
                r <<= 1
            return r
        rec_wgs = 2 * -(-Bk // (8 if form_used == 2 else 16)) if wide_used else 2 * -(-Bk // rows_per_block(Bk))
        for k in per_slab:
            if k.startswith("lstm_rec"):
                wgs[k] = rec_wgs
        cu_ms = {k: v * min(1.0, wgs.get(k, 256) / 256.0) for k, v in per_slab.items()}
        order = sorted(cu_ms, key=cu_ms.get, reverse=True)

        pmc = {}
        for fn in ("hbm_traffic.json", "mfma_util.json"):
            path = os.path.join(ROOT, "profiles", fn)
            if os.path.exists(path):
                with open(path) as f:
                    pmc[fn] = json.load(f)

        def roof_of(name, avg_ms, launches, where):
            by = algorithmic_bytes(name, Bk, T_r, T_e, W, S)
            if by:
                achieved = by / (avg_ms * 1e-3) / 1e9
                r = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(achieved / PEAK_HBM_GBS, 4), "avg_launch_ms": round(avg_ms, 5), "launches": launches,
                     "bytes_per_launch": by, "traffic": None}
            elif name.startswith("gemm_") and not args.per_step_decode:
                # split-operand GEMMs: every fp32 product is three exact f16 part products on v_mfma_f32_16x16x32_f16 -- priced as the
                # MFMA operations they issue (3 x the algorithmic FLOPs) against the dense f16 MFMA peak
                fl = algorithmic_flops(name, Bk, T_r, T_e, W, S, wide=wide_used)
                achieved = 3 * fl / (avg_ms * 1e-3) / 1e12
                r = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4), "avg_launch_ms": round(avg_ms, 5), "launches": launches,
                     "flops_per_launch": fl, "mfma_ops_per_launch": 3 * fl, "traffic": None,
                     "peak_note": "dense f16 MFMA peak; achieved = 3 x algorithmic FLOPs (three exact f16 part products per fp32 product) / launch time"}
            else:
                fl_ref = algorithmic_flops(name, Bk, T_r, T_e, W, S, wide=wide_used)
                fl = algorithmic_flops(name, Bk, T_r, T_e, W, S, chunk_steps, wide=wide_used) if fl_ref else None
                achieved = fl / (avg_ms * 1e-3) / 1e12 if fl else None
                r = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3) if achieved else None,
                     "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F32_TFLOPS, 4) if achieved else None,
                     "avg_launch_ms": round(avg_ms, 5), "launches": launches, "flops_per_launch": fl,
                     "flops_reference": fl_ref, "flops_executed": fl, "traffic": None}
            r["measured"] = where
            # rocprof PMC figures of the same kernel, collected on synchronous calls of this workload (tools/collect_traffic.sh,
            # tools/pmc_mfma.sh -> profiles/): HBM bytes per launch -> GB/s against the 8 TB/s peak; matrix-pipe busy fraction
            key = ("wide:" if wide_used else "fma:") + name
            t = pmc.get("hbm_traffic.json", {}).get(key)
            if t and Bk == 256:
                r["traffic"] = t["hbm_bytes_per_launch"]
                iso = per_slab.get(name, avg_ms)
                r["hbm_gbps"] = round(t["hbm_bytes_per_launch"] / (iso * 1e-3) / 1e9, 1)
                r["hbm_frac_of_peak"] = round(r["hbm_gbps"] / PEAK_HBM_GBS, 4)
                r["hbm_note"] = "PMC bytes per launch / the launch's duration alone on the chip"
            m = pmc.get("mfma_util.json", {}).get(key)
            if m and Bk == 256:
                r["mfma_util"] = m["mfma_util"]
                r["mfma_util_note"] = m.get("note", "")
            return r

        name = order[0]
        # `roofline` = the dominant kernel's launch ALONE on the chip (hipEvents on the library's stream around it, an untimed pass of
        # synchronous calls of the timed region's kernels right after the timed region): a kernel figure, and the contract's cross-check
        # -- the dominant kernel's time per step <= ms_per_step -- holds for it.  The same launch timed INSIDE the timed region
        # (profile 3) spans the other slabs' kernels that share the chip with it, so it is reported as a span, not as a roofline.
        roof = roof_of(name, per_slab[name], 10, "hipEvents on the library's stream around the launch, the launch alone on the chip "
                                                  "(untimed pass of 10 synchronous calls after the timed region)")
        roof["share_of_cu_time"] = round(cu_ms[name] / sum(cu_ms.values()), 3)
        if roof["bound"] == "mfma" and "peak_note" not in roof:
            roof["peak_note"] = PEAK_NOTE
        if name in prof_dec and roof.get("flops_per_launch"):
            ms, n = prof_dec[name]
            span = ms / max(n, 1)
            roof["span_ms_in_stream"] = round(span, 5)
            roof["launches_in_stream"] = n
            roof["frac_live_span"] = round(roof["flops_per_launch"] / (span * 1e-3) / 1e12 / roof["peak"], 4)
            chip_ms = (dt / args.steps) * 1e3 / max(slabs_per_step, 1) * roof["share_of_cu_time"]
            roof["chip_time_ms_per_launch"] = round(chip_ms, 5)
            roof["frac_chip_time"] = round(roof["flops_per_launch"] / (chip_ms * 1e-3) / 1e12 / roof["peak"], 4)
            roof["span_note"] = (f"span_ms_in_stream = hipEvents around the same launch inside an untimed copy of the timed region's stream, {max(depth, 1)} slab(s) in flight: "
                                 "other slabs' kernels run on the same CUs inside it (frac_live_span prices queueing, not the kernel); "
                                 "frac_chip_time = FLOPs / (timed ms per slab x share_of_cu_time): what the kernel costs the chip in the stream")
        top2 = [roof_of(k, per_slab[k], 10, "hipEvents, untimed pass, the launch alone on the chip") for k in order[1:3]
                if algorithmic_flops(k, Bk, T_r, T_e, W, S)]
        for r, k in zip(top2, [k for k in order[1:3] if algorithmic_flops(k, Bk, T_r, T_e, W, S)]):
            r["share_of_cu_time"] = round(cu_ms[k] / sum(cu_ms.values()), 3)
        # whole-path view: algorithmic FLOPs of the path / step time vs the fp32 peak.  `flops_reference`
        # credits the decode with B*W*S row-steps (the reference's slab-wide loop); `flops_executed` with the chunk-steps
        # the persistent decode really ran on this rank's first slab (a chunk leaves the loop once its beams are finished).
        enc_fl = T_r * 1050624 + T_e * 1058816 + (T_r + T_e) * 65536
        path_ref = enc_fl + W * S * DEC_FLOPS(T_r + T_e)
        exec_steps = float(np.mean(chunk_steps)) if chunk_steps is not None else float(S)
        path_exec = enc_fl + W * exec_steps * DEC_FLOPS(T_r + T_e)
        tf_ref = n_global * path_ref / (dt / args.steps) / 1e12
        tf_exec = n_global * path_exec / (dt / args.steps) / 1e12
        mode = "strong" if args.strong else "weak"
        out = {
            "metric": "kbases/s, raw+event joint mode, beam=5 (hot path: beam_search_prediction)",
            "value": round(chunks_per_s * BASES_PER_CHUNK / 1000.0, 3), "unit": "kbases/s",
            "value_note": "nominal: chunks/s x 6 bases per chunk (stride 6); random-init weights call mostly empty strings -- see read_level",
            "chunks_per_s": round(chunks_per_s, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": mode,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3: joint raw+event, T_raw={T_r}, T_event={T_e}, beam={W}, "
                                   + (f"{B} chunks/GPU/step" if not args.strong else f"one read of {n_global} chunks/step in slabs of {B}")
                                   + f", max_output_len={L}, enc_depth=2, dec_depth=1, units=128, luong",
                       "decode_steps": S, "decode_steps_executed_mean": round(exec_steps, 2),
                       "weights": "random-init (Keras defaults, seed 22)",
                       "pipelining": (f"the {args.steps} timed steps stream through the asynchronous calls (rv_beam_search_submit_dev / collect_dev), "
                                      f"{depth} slabs in flight, every step collected inside the timed region; results byte-identical to the synchronous call"
                                      if depth else "synchronous calls, one slab at a time"),
                       "recurrence": (f"matrix pipe, {8 if form_used == 2 else 16} chunks per workgroup (k_lstm_rec_mx + split-f16 projection GEMM)" if wide_used
                                      else "packed fp32 FMAs (k_lstm_rec_tw / k_lstm_rec_proj)"),
                       "parallelism": f"chunk-shard x{world}" + ((" + 1 RCCL all-gather/step (dist.sharded_beam_search_stream)" if (args.gather_per_step or not depth) else
                                                                       " + ONE RCCL all-gather at the end of the steps (dist.sharded_beam_search_many)") if dist_path else "")},
            "roofline": roof,
            "roofline_top2": top2,
            "roofline_path": {"bound": "mfma", "achieved": round(tf_exec, 2), "peak": PEAK_F32_TFLOPS * world, "unit": "TFLOP/s",
                              "frac": round(tf_exec / (PEAK_F32_TFLOPS * world), 4),
                              "flops_per_chunk_reference": path_ref, "flops_per_chunk_executed": round(path_exec),
                              "achieved_reference": round(tf_ref, 2), "frac_reference": round(tf_ref / (PEAK_F32_TFLOPS * world), 4),
                              "peak_note": PEAK_NOTE},
            "kernel_ms_per_launch_alone": {k: round(v, 4) for k, v in sorted(per_slab.items())},
            "kernel_cu_ms_per_slab": {k: round(v, 4) for k, v in sorted(cu_ms.items())},
            "decode_kernel_ms_per_launch": {k: round(v[0] / max(v[1], 1), 5) for k, v in sorted(dec.items())},
            # (streamed steps complete in bursts: these are intervals between step COMPLETIONS, not step latencies)
            # (one gather for all K steps: every step completes at the same instant -- no intervals to report)
            "step_completion_interval_ms_min_med_max": (None if (not plain_steps and depth and not args.gather_per_step) else
                                                        [round(x * 1e3, 3) for x in (min(per_step), sorted(per_step)[len(per_step) // 2], max(per_step))]),
            "verified": verified,
        }
        if gather_step_ms is not None:
            out["gather_per_step"] = {"ms_per_step": round(gather_step_ms * 1e3, 4), "chunks_per_s": round(n_global / gather_step_ms, 1),
                                      "note": "untimed extra: the same steps with one all-gather per step (dist.sharded_beam_search_stream) -- the protocol of "
                                              "earlier rounds' multi-GPU lines; the headline gathers ONCE for all K steps (dist.sharded_beam_search_many)"}
        if steady is not None:
            out["steady_state"] = {"ms_per_step": round(steady * 1e3, 4), "chunks_per_s": round(n_global / steady, 1),
                                   "note": f"untimed extra: {4 * max(args.steps, 10)} more steps of the same stream right after another {max(args.steps, 10)}, "
                                           "fill and drain amortised, clocks settled"}
        if sync_ms is not None:
            out["synchronous"] = {"ms_per_step": round(sync_ms, 4), "chunks_per_s": round(Bk / sync_ms * 1e3, 1),
                                  "note": "rv_beam_search_dev, one slab at a time, the SAME kernels as the headline (recurrence form "
                                          f"'{args.recurrence}'); untimed extra: 20 calls after the timed region"}
            out["synchronous_auto"] = {"ms_per_step": round(sync_auto_ms, 4), "chunks_per_s": round(Bk / sync_auto_ms * 1e3, 1),
                                       "note": "the same with wide_recurrence = -1 (form chosen per call: for one isolated slab of 160-512 chunks the matrix-pipe "
                                               "recurrence with EIGHT chunks per workgroup, the latency form -- other kernels than the headline's; results "
                                               "agree to f32 rounding)"}
    bc.close()
    if rank == 0:
        if world == 1 and not args.no_extras:      # sub-lines outside the timed region (own handles)
            try:
                out["read_level"] = read_level(rv, local)
                out["read_level_pipelined"] = read_level(rv, local, pipelined=True)
                out["read_level_concurrent"] = read_level(rv, local, pipelined=True, concurrent=4)
                out["bahdanau"] = variant_timing(rv, local, B, T_r, T_e, W, L, "Bahdanau attention", attention_type="bahdanau")
                out["enc3_dec2"] = variant_timing(rv, local, B, T_r, T_e, W, L, "encoder depth 3, two decoder cells, Luong attention",
                                                  encoder_depth=3, decoder_depth=2, attention_type="luong")
            except Exception as e:
                out["extras_error"] = repr(e)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(rv, bc.cfg, flat, T_r, T_e, W, L, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if dist_path:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
