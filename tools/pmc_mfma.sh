# usage: bash tools/pmc_mfma.sh  -- matrix-pipe and VALU busy counters of the fused recurrence+projection kernel and the GEMM
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "k_lstm_rec|k_gemm|k_dec_persist" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcM -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-pass > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections,os
root=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(f'{root}/gpurun_out/pmcM/*/*counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:60]+" grid "+r['Grid_Size']
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    print(k)
    for c,x in sorted(v.items()):
        print(f"   {c:30s} per-call {x/cnt[(k,c)]:16.0f}")
PY
