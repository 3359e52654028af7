# Timing ablation of k_dec_attend: RV_ATT_STOP=k makes the kernel return after phase k (results invalid).
# The switch is compiled only into the diagnostic library (`make -C ravvent-basecaller_amd/csrc diag`): the product library ignores it.
export RAVVENT_HIP_LIB=${RAVVENT_HIP_LIB:-$GRAFT_REPO_ROOT/ravvent-basecaller_amd/csrc/libravvent_hip_diag.so}
[ -f "$RAVVENT_HIP_LIB" ] || { echo "attend_ablation.sh: $RAVVENT_HIP_LIB not found -- build it with: make -C ravvent-basecaller_amd/csrc diag" >&2; exit 1; }
cd /tmp && export TMPDIR=/tmp
for k in ${STOPS:-1 2 3 4 5 6 7 0}; do
  RV_ATT_STOP=$k rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl$k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras >/dev/null 2>&1
  python3 - $k <<'PY'
import csv,sys,glob,os
k=sys.argv[1]
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+f'/gpurun_out/abl{k}/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_dec_attend' in r['Name'] or 'k_dec_cell' in r['Name']: print('stop',k, r['Name'][27:60], r['Calls'], round(float(r['AverageNs'])/1e3,2),'us')
PY
done
