#!/bin/bash
# experiment builds go to m3ae_amd/lib_diag/ (m3ae_amd/build.py) and are loaded with M3AE_DIAGNOSTIC_LIB=1: the product library is never touched
# timing of the per-head operand-build kernel (csrc/xattn.hip: xbuild_kernel) at a few tile shapes: rebuild with each and time
# both directions of the fused cross-attention forward; the LAST build is the default shape again
for shape in "128,256,2,2" "128,384,2,1" ""; do
    touch mm-vqa-healthcare_amd/csrc/xattn.hip
    flags=""; [ -n "$shape" ] && flags="-DM3AE_XBUILD_SHAPE=$shape"
    (cd mm-vqa-healthcare_amd && M3AE_EXTRA_HIPCC_FLAGS="$flags" python -m m3ae_amd.build > /dev/null) || exit 1
    M3AE_DIAGNOSTIC_LIB=$([ -n "$flags" ] && echo 1) python tools/xf_time.py "xbuild shape: ${shape:-default}" 2>&1 | grep -v amdgpu.ids
done
