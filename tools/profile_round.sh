# One gpurun call: kernel-trace stats, PMC traffic + utilisation passes per precision, then the bench lines.
# usage: bash tools/profile_round.sh r2 ["fp16 bf16 fp8"]
set -e
TAG=${1:-r2}
PRECS=${2:-"fp16 bf16 fp8"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
for P in $PRECS; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$P -- python3 bench.py --precision $P --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/$TAG/prof_${P}_bench.log 2>&1
cp gpurun_out/prof_${TAG}_$P/*/*kernel_stats.csv gpurun_out/$TAG/${P}_kernel_stats.csv
echo "stats $P done"
rocprofv3 -i tools/pmc_traffic.txt --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$P -- python3 bench.py --precision $P --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs > gpurun_out/$TAG/pmc_$P.log 2>&1
python3 tools/pmc_summarize.py gpurun_out/pmc_${TAG}_$P gpurun_out/$TAG/${P}_pmc_traffic.json "rocprofv3 -i tools/pmc_traffic.txt --kernel-trace -- python3 bench.py --precision $P --steps 2 --warmup 1 --no-cpu-baseline --no-roofline" 16 $P > /dev/null
echo "pmc traffic $P done"
rocprofv3 -i tools/pmc_util.txt --kernel-trace --output-format csv -d gpurun_out/pmcu_${TAG}_$P -- python3 bench.py --precision $P --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs > gpurun_out/$TAG/pmcu_$P.log 2>&1
python3 tools/pmc_kernel.py gpurun_out/pmcu_${TAG}_$P gemm16v5_kernel gemm8_kernel attn_window_kernel attn_global8_kernel attn_global_kernel layernorm_tiled > gpurun_out/$TAG/${P}_pmc_util.txt
echo "pmc util $P done"
done
for P in $PRECS; do
python3 bench.py --precision $P --no-other-configs --no-cpu-baseline > gpurun_out/$TAG/bench_$P.json 2> gpurun_out/$TAG/bench_$P.err
tail -1 gpurun_out/$TAG/bench_$P.json | cut -c1-200
done
