# GPU run 5: fp8 head/tail sweep (accuracy + speed), fp8 kernel stats
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
for cfg in "0 0" "0 4" "0 8" "4 0" "4 4" "8 8"; do
set -- $cfg
export WM_FP8_BF16_HEAD=$1 WM_FP8_BF16_TAIL=$2
python3 -m pytest tests/test_gpu_e2e.py -q -s -k "vit_h_fp8" 2>&1 | grep "^\[vit_h/fp8\]\|passed\|failed" | cut -c1-200 > gpurun_out/r2e/fp8_h$1_t$2.txt || true
python3 bench.py --precision fp8 --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | grep -o '"value": [0-9.]*' >> gpurun_out/r2e/fp8_h$1_t$2.txt
echo "head $1 tail $2: $(cat gpurun_out/r2e/fp8_h$1_t$2.txt | tr '\n' ' ')"
done
unset WM_FP8_BF16_HEAD WM_FP8_BF16_TAIL
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2e_fp8 -- python3 bench.py --precision fp8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2e/prof_fp8_bench.log 2>&1
head -8 gpurun_out/prof_r2e_fp8/*/*kernel_stats.csv | cut -c1-160
