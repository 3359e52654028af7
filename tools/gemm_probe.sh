# timing probes of the split-f16 GEMMs (results invalid): RV_GEMM_DBG bit 1 = no C stores, 2 = no A loads after the first, 4 = no MFMAs
# needs the diagnostic build: make -C ravvent-basecaller_amd/csrc gemm_diag (the product library ignores RV_GEMM_DBG)
R=${GRAFT_REPO_ROOT:-.}
for d in 0 1 2 4 3 6 7; do
  RAVVENT_HIP_LIB=$R/ravvent-basecaller_amd/csrc/libravvent_hip_gemm_diag.so RV_GEMM_DBG=$d python $R/tools/gemm_probe.py 2>/dev/null
done
