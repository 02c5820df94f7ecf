#!/bin/bash
# round 4: attention in the log2 domain (q pre-scaled, biases through the matrix pipe): op tests, e2e suite, A/B against the previous build
set -o pipefail
O=gpurun_out/${1:-r4c}; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -x -k "attention or mha16 or window" > $O/ops.log 2>&1; echo "ops rc=$? $(tail -1 $O/ops.log)"
python -m pytest tests/test_gpu_e2e.py -q -s --maxfail=12 > $O/e2e.log 2>&1; echo "e2e rc=$? $(tail -1 $O/e2e.log)"
grep -E "^FAILED|^ERROR" $O/e2e.log
bash tools/ab_builds.sh ${1:-r4c}/ab "new=-" "pre=WM_HIP_LIB=build/ab/libwm_pre_attn.so"
