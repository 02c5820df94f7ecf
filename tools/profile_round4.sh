#!/bin/bash
# One gpurun call per precision list: rocprofv3 kernel-trace stats, the two PMC passes (traffic, utilisation), then the bench line.
# usage: bash tools/profile_round4.sh r4 "fp16 bf16"      (results under gpurun_out/<tag>/, copied into profiles/ by hand)
set -e
TAG=${1:-r4}
PRECS=${2:-"fp16 bf16 fp8"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
for P in $PRECS; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$P -- python3 bench.py --precision $P --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/$TAG/prof_${P}_bench.log 2>&1
cp gpurun_out/prof_${TAG}_$P/*/*kernel_stats.csv gpurun_out/$TAG/${P}_kernel_stats.csv
python3 tools/gemm_by_shape.py gpurun_out/prof_${TAG}_$P > gpurun_out/$TAG/${P}_gemm_by_shape.txt 2>&1 || true
echo "stats $P done"
rocprofv3 -i tools/pmc_traffic.txt --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$P -- python3 bench.py --precision $P --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs > gpurun_out/$TAG/pmc_$P.log 2>&1
python3 tools/pmc_summarize.py gpurun_out/pmc_${TAG}_$P gpurun_out/$TAG/${P}_pmc_traffic.json "rocprofv3 -i tools/pmc_traffic.txt --kernel-trace -- python3 bench.py --precision $P --steps 2 --warmup 1 --no-cpu-baseline --no-roofline" 16 $P > /dev/null
echo "pmc traffic $P done"
rocprofv3 -i tools/pmc_util.txt --kernel-trace --output-format csv -d gpurun_out/pmcu_${TAG}_$P -- python3 bench.py --precision $P --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs > gpurun_out/$TAG/pmcu_$P.log 2>&1
# all 16-bit GEMM launches, then per instance: FOLDC (qkv, lin1), SPLIT (proj, lin2); the attention kernels; what is left of the LayerNorm
python3 tools/pmc_kernel.py gpurun_out/pmcu_${TAG}_$P gemm16v5_kernel "3, false, false, true, false>" "3, false, true, false, true>" gemm8_kernel attn_window_kernel attn_global8_kernel attn_global_kernel layernorm_tiled layernorm_plane_fp8 ln_stats_x16 > gpurun_out/$TAG/${P}_pmc_util.txt
echo "pmc util $P done"
rm -rf gpurun_out/prof_${TAG}_$P gpurun_out/pmc_${TAG}_$P gpurun_out/pmcu_${TAG}_$P
done
for P in $PRECS; do
python3 bench.py --precision $P --no-other-configs --no-cpu-baseline > gpurun_out/$TAG/bench_$P.json 2> gpurun_out/$TAG/bench_$P.err
tail -1 gpurun_out/$TAG/bench_$P.json | cut -c1-200
done
