#!/bin/bash
# Dev: time the attention kernels of several library builds on ONE box, two rounds.
# usage: tools/attn_ab.sh <outdir> <lib | - (in-tree) | old (in-tree, WM_ATTN_4WAVE=1)> ...   (lib = build/ab/libwm_<lib>.so)
O=gpurun_out/$1; shift; mkdir -p $O
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then r=$(python tools/attn_bench.py --batch 16 --prec fp16 2>&1 | grep "global\|window" | tr '\n' ' ');
    elif [ "$lib" = "old" ]; then r=$(WM_ATTN_4WAVE=1 python tools/attn_bench.py --batch 16 --prec fp16 2>&1 | grep "global\|window" | tr '\n' ' ');
    else r=$(WM_HIP_LIB=build/ab/libwm_$lib.so python tools/attn_bench.py --batch 16 --prec fp16 2>&1 | grep "global\|window" | tr '\n' ' '); fi
    echo "$lib round $round: $r" | tee -a $O/ab.txt
  done
done
