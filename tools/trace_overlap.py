"""Analyse a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W` (asynchronous calls): the TIMED REGION is the span of
decode launches W .. W+K-1 (the run also holds the warm-up, an isolated per-kernel pass and a synchronous pass); per-kernel launch
spans inside it, how many kernels are in flight, and the workgroup demand against the chip's 256 CUs.
Usage: trace_overlap.py <kernel_trace.csv> [K] [W] [depth]   (depth given = the round-4 bench: the timed region is followed by an untimed
streamed pass of depth + K steps with events around the decode, then the isolated pass of synchronous calls)"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20     # the driver's command: --steps 20 --warmup 5
W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
D = int(sys.argv[4]) if len(sys.argv) > 4 else None
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
    name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:30]
    wg = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) * (int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1))
    ev.append((s, e, name, wg))
ev.sort()
dec = [x for x in ev if x[2].startswith("k_dec_persist")]
tl = dec[W:W + K]
lo, hi = tl[0][0] - 1_500_000, tl[-1][1]
sel = [x for x in ev if x[0] >= lo and x[1] <= hi + 200_000]
span = (hi - lo) / 1e6
print(f"{len(dec)} decode launches in the trace; timed region = launches {W}..{W + K - 1}: {span:.2f} ms -> {span / K:.3f} ms per slab")
by = collections.defaultdict(list)
for s, e, n, w in sel:
    by[(n, w)].append((e - s) / 1e6)
print("launch spans inside the timed region (other slabs' kernels share the chip during a span):")
for (n, w), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n[:44]:44s} workgroups {w:5d}  n={len(v):3d}  avg {sum(v) / len(v):.4f} ms  min {min(v):.4f}  max {max(v):.4f}")
pts = []
for s, e, n, w in sel:
    pts.append((s, 1, min(w, 256))); pts.append((e, -1, -min(w, 256)))
pts.sort()
cur = dem = 0; last = pts[0][0]; hist = collections.Counter(); dsum = 0.0
for t, d, w in pts:
    dt = (t - last) / 1e6; hist[cur] += dt; dsum += dt * min(dem, 256) / 256; cur += d; dem += w; last = t
print("kernels in flight -> ms:", {k: round(v, 2) for k, v in sorted(hist.items())})
print(f"workgroup demand (each launch capped at 256 CUs, sum capped at the chip) averages {100 * dsum / span:.0f} % of the chip")
iso = [x for x in (dec[W + K:W + K + 10] if D is None else dec[W + K + D + K:W + K + D + K + 10])]
if iso:
    print(f"isolated pass (synchronous calls, the launch alone): k_dec_persist avg {sum((e - s) for s, e, _, _ in iso) / len(iso) / 1e6:.4f} ms over {len(iso)} launches")
