#!/bin/bash
# experiment builds go to m3ae_amd/lib_diag/ (m3ae_amd/build.py) and are loaded with M3AE_DIAGNOSTIC_LIB=1: the product library is never touched
# timing-only experiments on the one-launch image-query kernel: rebuild csrc/xflash.hip with each flag set and time it
for flags in "" "-DXF_EXP_NO_X" "-DXF_EXP_NO_KP" "-DXF_EXP_NO_VP" "-DXF_EXP_NO_X -DXF_EXP_NO_KP -DXF_EXP_NO_VP"; do
    touch mm-vqa-healthcare_amd/csrc/xflash.hip
    (cd mm-vqa-healthcare_amd && M3AE_EXTRA_HIPCC_FLAGS="$flags" python -m m3ae_amd.build > /dev/null) || exit 1
    M3AE_DIAGNOSTIC_LIB=$([ -n "$flags" ] && echo 1) python tools/xf_time.py "flags: $flags" 2>&1 | grep -A3 "img<-txt"
done
