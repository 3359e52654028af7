# Round-4 profile collection (run on the GPU box from the repo root: bash tools/collect_r04.sh).  Writes under gpurun_out/r04/:
#   kernel_stats.csv / kernel_trace.csv / bench_under_rocprof.json : rocprofv3 --kernel-trace --stats of the driver's bench command
#   hbm_traffic.json, mfma_util.json, pmc_mfma.txt : PMC passes (separate runs, --pmc with --kernel-trace only) over SYNCHRONOUS
#     calls (bench.py --depth 0), so that every launch is alone on the chip, for both recurrence forms
#   pmc_l2_decode.txt : L2 request / hit counters of the decode launch (the cell product's weight stream)
# Copy what should be judged into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv; cp $O/trace/*/*kernel_trace.csv $O/kernel_trace.csv
echo "trace done"
for form in mx fma; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${form}_$c -- python3 $R/bench.py --steps 3 --warmup 1 --depth 0 --recurrence $form --no-cpu-baseline --no-extras > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_${form}_MFMA -- python3 $R/bench.py --steps 3 --warmup 1 --depth 0 --recurrence $form --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "pmc $form done"
done
# L2 (TCC) traffic of the decode launch: requests, hits and misses, and the vector L1's read requests to the L2
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-include-regex "k_dec_persist" --kernel-trace --output-format csv -d $O/pmc_l2 -- python3 $R/bench.py --steps 3 --warmup 1 --depth 0 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_l2.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/pmc_l2/*/*counter_collection.csv")
out = open("$O/pmc_l2_decode.txt", "w")
if not f:
    out.write("no counter_collection.csv (counter names not available?)\n")
else:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(agg.items()):
        out.write(f"{c:28s} per decode launch {sum(v) / len(v):16.0f}   ({len(v)} launches)\n")
out.close()
print(open("$O/pmc_l2_decode.txt").read())
PY
cd $R && python3 tools/parse_r03.py $O && python3 tools/trace_overlap.py $O/kernel_trace.csv 20 5 10 > $O/trace_overlap.txt; cat $O/trace_overlap.txt
