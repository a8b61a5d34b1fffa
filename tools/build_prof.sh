#!/bin/bash
# Diagnostic build of the HIP library with in-kernel s_memtime stamps (-DRT_PROF).
# Select it with RT_AMD_LIBNAME=librt_amd_prof.so; never quote its run time, only its shares.
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DRT_PROF \
    -o ../lib/librt_amd_prof.so rt_capi.hip
