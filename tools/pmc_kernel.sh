# usage: KREGEX=k_dec_attend bash tools/pmc_kernel.sh  -- SQ counters for matching kernels (one pass)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-include-regex "${KREGEX:-k_dec}" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcK -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-include-regex "${KREGEX:-k_dec}" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcK2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections,os
root=os.environ['GRAFT_REPO_ROOT']
for d in ('pmcK','pmcK2'):
    f=glob.glob(f'{root}/gpurun_out/{d}/*/*counter_collection.csv')[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); grid={}
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:70]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        cnt[(k,r['Counter_Name'])]+=1
        grid[k]=(r['Grid_Size'],r['Workgroup_Size'],r.get('VGPR_Count'),r.get('Scratch_Size'),r.get('LDS_Block_Size'))
    for k,v in agg.items():
        print(k, grid[k])
        for c,x in sorted(v.items()):
            n=cnt[(k,c)]; print(f"   {c:22s} per-call {x/n:14.0f}")
PY
