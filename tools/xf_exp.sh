#!/bin/bash
# timing-only experiments on the one-launch image-query kernel: rebuild csrc/xflash.hip with each flag set and time it
for flags in "" "-DXF_EXP_NO_X" "-DXF_EXP_NO_KP" "-DXF_EXP_NO_VP" "-DXF_EXP_NO_X -DXF_EXP_NO_KP -DXF_EXP_NO_VP"; do
    touch mm-vqa-healthcare_amd/csrc/xflash.hip
    (cd mm-vqa-healthcare_amd && M3AE_EXTRA_HIPCC_FLAGS="$flags" python -m m3ae_amd.build > /dev/null) || exit 1
    python tools/xf_time.py "flags: $flags" 2>&1 | grep -A3 "img<-txt"
done
