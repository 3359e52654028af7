"""Analyse a rocprofv3 --kernel-trace CSV of bench.py: per-kernel launch spans, how many kernels run at once, and how much of the
wall time the GPU has at least one / the decode kernel running.  Usage: trace_overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "")
    wgs = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) * max(int(r.get("Grid_Size_Y", 1) or 1) // max(int(r.get("Workgroup_Size_Y", 1) or 1), 1), 1)
    ev.append((s, e, name, wgs))
ev.sort()
# timed region = the last 60 % of the trace (skip warm-up / set-up)
t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)
lo = t0 + int(0.4 * (t1 - t0))
sel = [x for x in ev if x[0] >= lo]
span = (max(e for _, e, _, _ in sel) - min(s for s, _, _, _ in sel)) / 1e6
by = collections.defaultdict(list)
for s, e, n, w in sel:
    by[n].append((e - s) / 1e6)
print(f"window {span:.2f} ms, {len(sel)} dispatches")
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n[:60]:60s} n={len(v):4d} avg {sum(v)/len(v):.4f} ms  total {sum(v):.2f} ms  ({100*sum(v)/span:.0f} % of window)")
# concurrency profile
pts = []
for s, e, n, w in sel:
    pts.append((s, 1, min(w, 256))); pts.append((e, -1, -min(w, 256)))
pts.sort()
cur = dem = 0; last = pts[0][0]; hist = collections.Counter(); demand_time = 0.0; busy = 0.0
for t, d, w in pts:
    dt = (t - last) / 1e6
    hist[cur] += dt
    demand_time += dt * min(dem, 256) / 256.0
    if cur > 0: busy += dt
    cur += d; dem += w; last = t
print("kernels in flight -> ms:", {k: round(v, 2) for k, v in sorted(hist.items())})
print(f"some kernel running {100*busy/span:.0f} % of the window; workgroup demand (capped at 256 CUs) averages {100*demand_time/span:.0f} % of the chip")
