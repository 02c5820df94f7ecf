#!/bin/bash
# per-shape GEMM durations in the model, split stream vs fp32 stream, one box (rocprofv3 kernel trace of bench.py --steps 4)
O=gpurun_out/${1:-r4shapes}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in split:1 fp32stream:0 split2:1 fp32stream2:0; do
  label=${spec%%:*}; v=${spec#*:}
  WM_STREAM_SPLIT=$v rocprofv3 --kernel-trace --output-format csv -d $O/prof_$label -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs > $O/$label.log 2>&1
  echo "== $label"; python3 tools/gemm_by_shape.py $O/prof_$label | grep -E "FOLDC|SPLIT|FOLDP"
  rm -rf $O/prof_$label
done
