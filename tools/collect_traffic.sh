# HBM traffic per launch from the L2's memory-side counters (MI355X_MICROARCH.md "HBM"): separate
# --pmc passes for FETCH_SIZE and WRITE_SIZE; on gfx950 FETCH_SIZE reports half the bytes of wide
# coalesced reads, so it is doubled.  Writes gpurun_out/hbm_traffic.json (copy to profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcF -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcW -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, json, os
root = os.environ["GRAFT_REPO_ROOT"]
names = {"k_dec_persist": "dec_persist", "k_dec_attend_flash": "dec_attend", "k_dec_attend<": "dec_attend_two_pass", "k_dec_cell": "dec_cell",
         "k_gemm_f32<2, 2": "gemm_memory", "k_gemm_mem_split": "gemm_memory", "k_lstm_rec_proj<2": "lstm_rec_l1p_fused", "k_lstm_rec<2, 0": "lstm_rec_l1p", "k_lstm_rec<2, 1": "lstm_rec_raw_l0", "k_lstm_rec_tw<2, 1": "lstm_rec_raw_l0",
         "k_lstm_rec_tw<2, 5": "lstm_rec_event_l0",
         "k_lstm_rec<2, 5": "lstm_rec_event_l0"}
out = collections.defaultdict(lambda: {"fetch_kb": [], "write_kb": []})
for d, key in (("pmcF", "fetch_kb"), ("pmcW", "write_kb")):
    f = glob.glob(f"{root}/gpurun_out/{d}/*/*counter_collection.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
    seen = collections.Counter()
    for r in rows:
        for pat, nm in names.items():
            if pat in r["Kernel_Name"]:
                grid = r["Grid_Size"]
                if nm == "lstm_rec_l1p_fused":       # same grid for both encoders; the event encoder is launched first in every slab
                    nm = "lstm_rec_event_l1p" if seen[pat] % 2 == 0 else "lstm_rec_raw_l1p"
                    seen[pat] += 1
                out[f"{nm}@grid{grid}"][key].append(float(r["Counter_Value"]))
                break
res = {}
for k, v in sorted(out.items()):
    f = sum(v["fetch_kb"]) / max(len(v["fetch_kb"]), 1); w = sum(v["write_kb"]) / max(len(v["write_kb"]), 1)
    res[k] = {"launches_sampled": len(v["fetch_kb"]), "FETCH_SIZE_KB_raw": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
              "hbm_bytes_per_launch": int((2 * f + w) * 1024),
              "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 128-B requests as 64 B"}
json.dump(res, open(f"{root}/gpurun_out/hbm_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
