#!/bin/bash
# round 4: op tests of the split stream, the e2e suite, the bf16 fold study, split vs fp32-stream A/B -- one gpurun call
set -o pipefail
O=gpurun_out/${1:-r4b}; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -x -k "split or stats or fold or unpack" > $O/ops.log 2>&1; echo "ops rc=$? $(tail -1 $O/ops.log)"
python -m pytest tests/test_gpu_e2e.py -q -s --maxfail=12 > $O/e2e.log 2>&1; echo "e2e rc=$? $(tail -1 $O/e2e.log)"
grep -E "^FAILED|^ERROR" $O/e2e.log
python tools/bf16_fold_study.py bf16 fp16 > $O/fold_study.txt 2>&1; echo "study rc=$?"; cat $O/fold_study.txt | grep seed
bash tools/ab_builds.sh ${1:-r4b}/ab "split=-" "fp32stream=WM_STREAM_SPLIT=0"
