#!/bin/bash
# experiment builds go to m3ae_amd/lib_diag/ (m3ae_amd/build.py) and are loaded with M3AE_DIAGNOSTIC_LIB=1: the product library is never touched
for flags in "-DM3AE_NT_TRACE" "-DM3AE_NT_TRACE -DM3AE_EXP_NT_CONTIG" "-DM3AE_NT_TRACE -DM3AE_EXP_NT_NODMA"; do
    touch mm-vqa-healthcare_amd/csrc/gemm_mfma.hip
    (cd mm-vqa-healthcare_amd && M3AE_EXTRA_HIPCC_FLAGS="$flags" python -m m3ae_amd.build > /dev/null) || exit 1
    M3AE_DIAGNOSTIC_LIB=$([ -n "$flags" ] && echo 1) python tools/nt_trace.py "$flags" 2>&1 | grep -A2 "^\["
done
