#!/bin/bash
# Dev build of the library with the in-kernel timeline instances (WM_GEMM_DBG / WM_GEMM8_DBG / WM_LNF_TIMELINE) and the GEMM
# timing bits of tools/gemm_bench.py (--act 256 / 512 / 1024).  Writes build/ab/libwm_dev.so (git-ignored, travels with gpurun);
# use it with WM_HIP_LIB=build/ab/libwm_dev.so.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DWM_DEV_TIMELINE=1 -DWM_GEMM_TIMING_BITS=1 \
    -I include -o build/ab/libwm_dev.so wildlifemapper_amd/csrc/wm_api.hip
