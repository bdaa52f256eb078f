#!/bin/bash
# experiment builds go to m3ae_amd/lib_diag/ (m3ae_amd/build.py) and are loaded with M3AE_DIAGNOSTIC_LIB=1: the product library is never touched
# timing-only experiments on the ping-pong NT GEMM: rebuild csrc/gemm_mfma.hip with each flag set and time the step's shapes
# usage: tools/nt_exp.sh "<flags 1>" "<flags 2>" ...   (default: the round-3 set)
if [ $# -eq 0 ]; then set -- "" "-DM3AE_EXP_NT_NODMA" "-DM3AE_EXP_NT_CONTIG" "-DM3AE_EXP_NT_NOSTORE" "-DM3AE_EXP_NT_NODMA -DM3AE_EXP_NT_NOSTORE" "-DM3AE_EXP_NT_L2HOT"; fi
for flags in "$@"; do
    touch mm-vqa-healthcare_amd/csrc/gemm_mfma.hip
    (cd mm-vqa-healthcare_amd && M3AE_EXTRA_HIPCC_FLAGS="$flags" python -m m3ae_amd.build > /dev/null) || exit 1
    M3AE_DIAGNOSTIC_LIB=$([ -n "$flags" ] && echo 1) python tools/nt_exp.py "flags: $flags" 2>&1 | grep "^\["
done
