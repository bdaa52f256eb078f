#!/bin/bash
for flags in "-DM3AE_NT_TRACE" "-DM3AE_NT_TRACE -DM3AE_EXP_NT_CONTIG" "-DM3AE_NT_TRACE -DM3AE_EXP_NT_NODMA"; do
    touch mm-vqa-healthcare_amd/csrc/gemm_mfma.hip
    (cd mm-vqa-healthcare_amd && M3AE_EXTRA_HIPCC_FLAGS="$flags" python -m m3ae_amd.build > /dev/null) || exit 1
    python tools/nt_trace.py "$flags" 2>&1 | grep -A2 "^\["
done
