# GPU run 2 of round 2: B=16 profile (stats + PMC traffic), LN-fusion A/B, fp16-tail sweep
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
bash tools/profile_round.sh r2b
python3 tools/pmc_summarize.py gpurun_out/pmc_r2b gpurun_out/r2b/pmc_traffic.json "rocprofv3 -i tools/pmc_traffic.txt --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline" 16 bf16 > gpurun_out/r2b/pmc_traffic.txt
echo "== LN fuse A/B"
for i in 1 2; do
python3 bench.py --no-cpu-baseline --steps 10 > gpurun_out/r2b/bench_plain_$i.json 2>/dev/null
WM_LN_FUSE=1 python3 bench.py --no-cpu-baseline --steps 10 > gpurun_out/r2b/bench_lnfuse_$i.json 2>/dev/null
done
grep -o '"value": [0-9.]*' gpurun_out/r2b/bench_plain_*.json gpurun_out/r2b/bench_lnfuse_*.json
echo "== fp16 tail sweep"
for t in 0 8 16 32; do
WM_FP16_TAIL=$t python3 -m pytest tests/test_gpu_e2e.py -q -s -k "vit_h_vs_reference_golden and bf16" 2>&1 | grep "logits margin\|passed\|failed" > gpurun_out/r2b/tail_$t.txt || true
WM_FP16_TAIL=$t python3 bench.py --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | grep -o '"value": [0-9.]*' >> gpurun_out/r2b/tail_$t.txt
echo "tail $t: $(cat gpurun_out/r2b/tail_$t.txt | tr '\n' ' ')"
done
