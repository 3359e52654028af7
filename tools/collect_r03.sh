# Round-3 profile collection (run on the GPU box from the repo root: bash tools/collect_r03.sh).  Writes under gpurun_out/r03/:
#   kernel_stats.csv / kernel_trace.csv / bench_under_rocprof.json : rocprofv3 --kernel-trace --stats of the DEFAULT bench command
#   hbm_traffic.json, mfma_util.json, pmc_mfma.txt : PMC passes (separate runs, --pmc with --kernel-trace only) over SYNCHRONOUS
#     calls (bench.py --depth 0), so that every launch is alone on the chip, for both recurrence forms
# Copy what should be judged into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
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
cd $R && python3 tools/parse_r03.py $O
