# One gpurun call: kernel-trace stats, PMC traffic passes, then the default bench line.  usage: bash tools/profile_round.sh r1f
set -e
TAG=${1:-r1x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${TAG}_bench.log 2>&1
echo "stats done"
rocprofv3 -i tools/pmc_traffic.txt --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_${TAG}.log 2>&1
echo "pmc done"
python3 bench.py > gpurun_out/bench_${TAG}_default.json 2> gpurun_out/bench_${TAG}_default.err
tail -1 gpurun_out/bench_${TAG}_default.json | cut -c1-300
