#!/bin/bash
# round-3 baseline on one box: bench at B = 16 / 8 / 4 (fp16), GEMM microbench incl. the guide's calibration shapes
set -o pipefail
O=gpurun_out/r3a; mkdir -p $O
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b16.json 2> $O/bench.err || exit 1
echo "b16 done"; tail -c 300 $O/bench_b16.json
python bench.py --steps 10 --warmup 3 --batch 8 --no-cpu-baseline > $O/bench_b8.json 2>> $O/bench.err || exit 1
python bench.py --steps 10 --warmup 3 --batch 4 --no-cpu-baseline > $O/bench_b4.json 2>> $O/bench.err || exit 1
echo "benches done"
{
python tools/gemm_bench.py --batch 16 --prec fp16
python tools/gemm_bench.py --batch 8 --prec fp16
python tools/gemm_bench.py --batch 16 --prec fp16 --residual --shapes proj,lin2
python tools/gemm_bench.py --batch 8 --prec fp16 --residual --shapes proj,lin2
python tools/gemm_bench.py --batch 4 --prec fp16 --residual --shapes proj,lin2
python tools/gemm_bench.py --prec bf16 --custom "4096,4096,4096;8192,8192,8192;16384,3840,4096;65536,1280,1280;65536,1024,1280"
} > $O/gemm.txt 2>&1
cat $O/gemm.txt
