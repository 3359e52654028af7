#!/usr/bin/env python3
"""bench.py -- throughput of the Ravvent hot path (Basecaller.beam_search_prediction) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one slab of synthetic chunks per GPU: workload C3 of
BASELINE.json (joint raw+event mode, 300-sample raw windows + 30 events, beam 5, 256 chunks,
max_output_len 48), inputs resident in HBM before the timed region, outputs left in HBM; with
N>1 every rank decodes its own slab (weak scaling, chunks shard embarrassingly) and one RCCL
all-gather of the [B,L-1] tokens+scores closes each step.  Rank 0 prints ONE JSON line.

metric: kbases/s = chunks/s * 6 bases per chunk / 1000 (stride 6 events => ~6 new bases per chunk,
/root/reference/ravvent_performance_evaluator.py:16); chunks/s is reported beside it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BASES_PER_CHUNK = 6
PEAK_F32_TFLOPS = 157.3     # MI355X fp32 vector = fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0       # HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s achievable)


def algorithmic_flops(kernel, B, T_r, T_e, W, S):
    """Algorithmic FLOPs of ONE launch of `kernel` (2 per MAC, pointwise ignored)."""
    Tm = T_r + T_e
    rec = 2 * 128 * 512 * 2            # recurrent product, both directions, per chunk-step
    return {
        "lstm_rec_raw_l0": B * T_r * (rec + 2 * 1 * 512 * 2),
        "lstm_rec_event_l0": B * T_e * (rec + 2 * 5 * 512 * 2),
        "lstm_rec_raw_l1p": B * T_r * (rec + 256 * 1024 * 2),      # recurrence + the fused input projection (MFMA waves)
        "lstm_rec_event_l1p": B * T_e * (rec + 256 * 1024 * 2),
        "gemm_inproj_raw": B * T_r * 256 * 1024 * 2,     # both directions per launch
        "gemm_inproj_event": B * T_e * 256 * 1024 * 2,
        "gemm_keys": B * Tm * 256 * 128 * 2,
        "gemm_memory": B * Tm * 256 * 256 * 2,               # [keys | attention-layer image of the values] for the persistent decode
        "decode_graph": B * W * S * (369408 + 768 * Tm),
        "dec_persist": B * W * S * (369408 + 768 * Tm),      # the whole decode loop is one launch
        "dec_cell": B * W * 2 * 256 * 512,
    }.get(kernel)


def algorithmic_bytes(kernel, B, T_r, T_e, W, S):
    """Algorithmic HBM bytes of ONE launch of an HBM-bound kernel."""
    Tm = T_r + T_e
    return {
        # single-pass Luong attend: the chunk's values [T_m,256] fp32 once per step (+ mask bytes)
        "dec_attend": B * Tm * (256 * 4 + 1),
    }.get(kernel)


def cpu_baseline(rv, cfg, flat, T_r, T_e, W, L, sample_chunks=0, target_s=12.0):
    """The C restatement of the oracle (oracle/ravvent_cpu.c, 'port') timed on this host's cores
    on a bounded sample of the same workload: a 256-chunk probe sizes the sample to ~target_s
    seconds of CPU work (sample_chunks > 0 fixes it instead)."""
    from oracle import cpu_port                       # checker / baseline only
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
    return {"value": round(n / dt * BASES_PER_CHUNK / 1000.0, 4), "unit": "kbases/s",
            "chunks_per_s": round(n / dt, 2), "cores": cores, "kind": "port",
            "sample": f"{n} chunks of the same workload (joint {T_r}+{T_e}, beam {W}, L {L}) in slabs of the "
                      f"C port's choosing, {dt:.1f} s wall on {cores} threads, S={tok.shape[1]}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="chunks per GPU per step (C3: 256)")
    ap.add_argument("--raw-len", type=int, default=300)
    ap.add_argument("--event-len", type=int, default=30)
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--max-output-len", type=int, default=48)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-step-decode", action="store_true",
                    help="A/B: per-step decode kernels in a hipGraph instead of the one-launch persistent decode")
    ap.add_argument("--attend-threads", type=int, default=0, help="0 auto | 256 | 512 (library option attend_threads)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL; the real path) | gloo (control-flow rehearsal: ranks may share a GPU, the gather goes through host memory)")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the per-kernel event pass over the decode loop")
    ap.add_argument("--cpu-sample", type=int, default=0, help="chunks for the CPU baseline (0 = size to ~12 s)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import ravvent_basecaller_amd as rv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.dist_backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
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
    if args.attend_threads:
        bc.set_option("attend_threads", args.attend_threads)
    if args.per_step_decode:
        bc.set_option("persistent_decode", 0)
    raw, ev, _ = rv.synthetic.make_slab(B, T_r, T_e, seed=rank)
    d_raw, d_ev = torch.from_numpy(raw).to(dev), torch.from_numpy(ev).to(dev)
    packed = gathered = None
    if world > 1:   # tokens (i32) and scores (f32 bits) travel in ONE fixed-shape all-gather per step
        packed = torch.zeros((B, 2 * (L - 1)), dtype=torch.int32, device=dev)
        gathered = torch.empty((world * B, 2 * (L - 1)), dtype=torch.int32, device=dev)

    def step():
        tok, sc = bc.beam_search_prediction((d_raw, d_ev), beam_width=W, max_output_len=L)
        if world > 1:   # the path's single exchange: gather every rank's calls (RCCL over xGMI)
            S = tok.shape[1]
            packed[:, :S] = tok
            packed[:, L - 1:L - 1 + S] = sc.view(torch.int32)
            if args.dist_backend == "nccl":
                dist.all_gather_into_tensor(gathered, packed)
            else:
                g_cpu = torch.empty(gathered.shape, dtype=gathered.dtype)
                dist.all_gather_into_tensor(g_cpu, packed.cpu())
        return tok, sc

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # hipEvents on the library's stream, live in the timed region, around the dominant launch (the decode) only:
    # events around all ~12 launches of a slab cost 2.6 % of it; the other kernels are timed in an untimed pass below
    bc.set_option("profile", 3)
    import gc
    gc.collect(); gc.disable()             # before the warm-up: a collection between warm-up and timing idles the GPU (~50 ms)
    for _ in range(max(args.warmup, 1)):   # (>=1: graph capture and event pool are built here, not in the timed region)
        step()
    bc.reset_profile()
    fence()
    t0 = time.perf_counter()
    per_step = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        tok, sc = step()
        per_step.append(time.perf_counter() - ts)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    S = int(tok.shape[1])
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof_dec = bc.profile()            # decode launch(es), measured inside the timed region
    prof = dict(prof_dec)
    if rank == 0:                      # every launch, in a short untimed pass (local decode only: no collective here)
        n_pass = min(args.steps, 10)
        bc.set_option("profile", 1)
        bc.reset_profile()
        for _ in range(n_pass):
            bc.beam_search_prediction((d_raw, d_ev), beam_width=W, max_output_len=L)
        prof = {k: (v[0] * args.steps / n_pass, v[1] * args.steps / n_pass) for k, v in bc.profile().items()}   # scaled to args.steps
        prof.update(prof_dec)
    if os.environ.get("RV_BENCH_VERBOSE"):
        print("per-step ms:", " ".join(f"{x*1e3:.2f}" for x in per_step), file=sys.stderr)

    if rank == 0:
        chunks_per_s = world * B * args.steps / dt
        # Per-kernel view of the decode graph: a short extra pass with hipEvents around EVERY kernel
        # (option profile=2: the graph is bypassed, same kernels, same stream).  Not part of `value`.
        dec = {}
        if not args.no_kernel_pass:
            bc.set_option("profile", 2)
            bc.reset_profile()
            for _ in range(3):     # local decode only: no collective here (this block runs on rank 0 alone)
                bc.beam_search_prediction((d_raw, d_ev), beam_width=W, max_output_len=L)
            dec = {k: v for k, v in bc.profile().items() if k.startswith("dec_") and k != "dec_finalize"}
            bc.set_option("profile", 1)
        # time per slab of every kernel name (graph replaced by its members when available)
        per_slab = {k: v[0] / args.steps for k, v in prof.items()}
        if dec:
            per_slab.pop("decode_graph", None)
            for k, (ms, n) in dec.items():
                per_slab[k] = ms / 3.0
        name = max(per_slab, key=per_slab.get)                    # dominant kernel
        # the decode launch is timed inside the timed region (profile 3); other kernels come from the untimed passes
        ms, n = prof_dec[name] if name in prof_dec else (dec[name] if name in dec else prof[name])
        avg_ms = ms / max(n, 1)
        by = algorithmic_bytes(name, B, T_r, T_e, W, S)
        if by:
            achieved = by / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBS, 4), "avg_launch_ms": round(avg_ms, 5), "launches": n,
                    "bytes_per_launch": by, "traffic": None}
        else:
            fl = algorithmic_flops(name, B, T_r, T_e, W, S)
            achieved = fl / (avg_ms * 1e-3) / 1e12 if fl else None
            roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3) if achieved else None,
                    "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_TFLOPS, 4) if achieved else None,
                    "avg_launch_ms": round(avg_ms, 5), "launches": n, "flops_per_launch": fl, "traffic": None}
        roof["share_of_slab_time"] = round(per_slab[name] / sum(per_slab.values()), 3)
        pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                for k, v in json.load(f).items():
                    if k.split("@")[0] == name and B == 256:          # collected on this workload only
                        roof["traffic"] = v["hbm_bytes_per_launch"]
        # whole-path view: algorithmic FLOPs of the path / step time vs the fp32 peak
        path_fl = B * (T_r * 1050624 + T_e * 1058816 + (T_r + T_e) * 65536 + W * S * (369408 + 768 * (T_r + T_e)))
        path_tf = world * path_fl / (dt / args.steps) / 1e12
        total_ms = sum(v[0] for v in prof.values())
        out = {
            "metric": "kbases/s, raw+event joint mode, beam=5 (hot path: beam_search_prediction)",
            "value": round(chunks_per_s * BASES_PER_CHUNK / 1000.0, 3), "unit": "kbases/s",
            "chunks_per_s": round(chunks_per_s, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3: joint raw+event, T_raw={T_r}, T_event={T_e}, beam={W}, "
                                   f"{B} chunks/GPU/step, max_output_len={L}, enc_depth=2, dec_depth=1, units=128, luong",
                       "decode_steps": S, "weights": "random-init (Keras defaults, seed 22)",
                       "parallelism": f"chunk-shard x{world}" + (" + 1 RCCL all-gather/step" if world > 1 else "")},
            "roofline": roof,
            "roofline_path": {"bound": "mfma", "achieved": round(path_tf, 2), "peak": PEAK_F32_TFLOPS * world, "unit": "TFLOP/s",
                              "frac": round(path_tf / (PEAK_F32_TFLOPS * world), 4), "flops_per_chunk": path_fl // B},
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(prof.items())},
            "decode_kernel_ms_per_launch": {k: round(v[0] / max(v[1], 1), 5) for k, v in sorted(dec.items())},
            "device_ms_per_step": round(total_ms / args.steps, 4),
            "step_ms_min_med_max": [round(x * 1e3, 3) for x in (min(per_step), sorted(per_step)[len(per_step) // 2], max(per_step))],
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(rv, bc.cfg, flat, T_r, T_e, W, L, args.cpu_sample)
        print(json.dumps(out), flush=True)
    bc.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
