# needs the diagnostic build: make -C ravvent-basecaller_amd/csrc gemm_diag (the product library ignores RV_GEMM_DBG)
# timing probes of the split-f16 memory GEMM (results invalid): RV_GEMM_DBG bit 1 = no C stores, 2 = no A loads after the first, 4 = no MFMAs
for d in 0 1 2 4 3 7; do
  echo -n "RV_GEMM_DBG=$d  gemm_memory ms per slab: "
  RAVVENT_HIP_LIB=${GRAFT_REPO_ROOT:-.}/ravvent-basecaller_amd/csrc/libravvent_hip_gemm_diag.so RV_GEMM_DBG=$d python ${GRAFT_REPO_ROOT:-.}/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys;print(json.loads(sys.stdin.read())['kernel_ms_per_slab']['gemm_memory'])"
done
