#!/bin/bash
# Dev build of the library with the in-kernel timeline instances (WM_GEMM_DBG / WM_GEMM8_DBG / WM_LNF_TIMELINE) and the GEMM
# timing bits of tools/gemm_bench.py (--act 256 / 512 / 1024).  Writes wildlifemapper_amd/libwm_hip.so in place: rebuild the
# product library afterwards (python -c "import __graft_entry__ as g; g.build()" after touching a source, or build.py --force).
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DWM_DEV_TIMELINE=1 -DWM_GEMM_TIMING_BITS=1 \
    -I include -o wildlifemapper_amd/libwm_hip.so wildlifemapper_amd/csrc/wm_api.hip
