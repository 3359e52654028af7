"""Turn the PMC passes of tools/collect_r03.sh into profiles-ready JSON: HBM bytes per launch (FETCH_SIZE doubled: gfx950 counts
128-byte requests as 64, MI355X_MICROARCH.md) and matrix-pipe utilisation per launch, keyed "<form>:<bench kernel name>"."""
import collections, csv, glob, json, os, re, sys
O = sys.argv[1]
CLOCK_GHZ = 2.4          # MI355X peak engine clock: utilisation = MFMA-busy cycles per SIMD / (launch duration x this clock), a LOWER bound


def slab_names(form):
    """Kernel launches of one synchronous C3 slab, in launch order -> bench.py's names."""
    if form == "mx":
        seq = [("k_lstm_rec_mx<5,", "lstm_rec_event_l0"),
               ("k_gemm_ws", "gemm_inproj_event"), ("k_lstm_rec_mx<0,", "lstm_rec_event_l1p"), ("k_lstm_rec_mx<1,", "lstm_rec_raw_l0"),
               ("k_gemm_ws", "gemm_inproj_raw"), ("k_lstm_rec_mx<0,", "lstm_rec_raw_l1p"), ("k_gemm_mem_split3", "gemm_memory"),
               ("k_dec_persist", "dec_persist"), ("k_dec_finalize", "dec_finalize")]
    else:
        seq = [("k_input_mask", "input_mask"), ("k_lstm_rec_tw<2, 5>", "lstm_rec_event_l0"), ("k_lstm_rec_proj", "lstm_rec_event_l1p"),
               ("k_lstm_rec_tw<2, 1>", "lstm_rec_raw_l0"), ("k_lstm_rec_proj", "lstm_rec_raw_l1p"), ("k_gemm_mem_split3", "gemm_memory"),
               ("k_dec_persist", "dec_persist"), ("k_dec_finalize", "dec_finalize")]
    return seq


def label(rows, form):
    """rows of one counter_collection.csv (one row per dispatch x counter) -> {dispatch id: bench name}: the mask kernel runs on a
    side stream, so launches are matched by kernel name in order of appearance within each kernel name's own sequence."""
    seq = slab_names(form)
    per_kernel = collections.defaultdict(list)
    for pat, nm in seq:
        per_kernel[pat].append(nm)
    seen = collections.Counter()
    out = {}
    for did, kname in sorted({(int(r["Dispatch_Id"]), r["Kernel_Name"]) for r in rows}):
        for pat, names in per_kernel.items():
            if pat in kname:
                out[did] = names[seen[pat] % len(names)]
                seen[pat] += 1
                break
    return out


def load(path):
    f = glob.glob(f"{path}/*/*counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []


traffic, util, txt = {}, {}, []
for form in ("mx", "fma"):
    key = {"mx": "wide", "fma": "fma"}[form]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = load(f"{O}/pmc_{form}_{c}")
        lab = label(rows, form)
        for r in rows:
            nm = lab.get(int(r["Dispatch_Id"]))
            if nm and r["Counter_Name"] == c:
                agg[nm][c].append(float(r["Counter_Value"]))
    for nm, v in agg.items():
        f = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1); w = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
        traffic[f"{key}:{nm}"] = {"launches_sampled": len(v["FETCH_SIZE"]), "FETCH_SIZE_KB_raw": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                                  "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                                  "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 128-B requests as 64 B"}
    rows = load(f"{O}/pmc_{form}_MFMA")
    lab = label(rows, form)
    tr = glob.glob(f"{O}/pmc_{form}_MFMA/*/*kernel_trace.csv")
    dur = {}
    if tr:
        for r in csv.DictReader(open(tr[0])):
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    m = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        nm = lab.get(int(r["Dispatch_Id"]))
        if nm:
            m[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
            m[nm]["ms"].append(dur.get(int(r["Dispatch_Id"]), 0.0))
    for nm, v in sorted(m.items()):
        avg = {c: sum(x) / len(x) for c, x in v.items()}
        txt.append(f"{key}:{nm}")
        for c, x in sorted(avg.items()):
            txt.append(f"   {c:28s} per launch {x:16.4f}")
        busy = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if busy > 0 and avg.get("ms", 0) > 0:
            per_simd = busy / (4 * 256)
            u = per_simd / (avg["ms"] * 1e-3 * CLOCK_GHZ * 1e9)
            util[f"{key}:{nm}"] = {"mfma_util": round(u, 4), "mfma_busy_cycles_per_simd": round(per_simd), "launch_ms_under_pmc": round(avg["ms"], 4),
                                   "mfma_instructions": round(avg.get("SQ_INSTS_MFMA", 0)),
                                   "note": f"SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) / (launch duration x {CLOCK_GHZ} GHz): the share of the launch the "
                                           "16-bit matrix pipe (2.5 PFLOP/s dense) is busy; synchronous calls, the launch alone on the chip"}
json.dump(traffic, open(f"{O}/hbm_traffic.json", "w"), indent=1)
json.dump(util, open(f"{O}/mfma_util.json", "w"), indent=1)
open(f"{O}/pmc_mfma.txt", "w").write("\n".join(txt) + "\n")
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in traffic.items()}, indent=1))
print(json.dumps({k: v["mfma_util"] for k, v in util.items()}, indent=1))
