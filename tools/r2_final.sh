# End-of-round validation: the whole GPU suite, smoke(), then the bench lines (default fp16 operands = configs[2]; bf16; fp8 = configs[4]).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2final
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r2final/pytest_gpu.txt 2>&1 || { tail -40 gpurun_out/r2final/pytest_gpu.txt; exit 1; }
tail -3 gpurun_out/r2final/pytest_gpu.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/r2final/smoke.txt 2>&1 || { tail -20 gpurun_out/r2final/smoke.txt; exit 1; }
tail -1 gpurun_out/r2final/smoke.txt
timeout -k 10 400 python bench.py > gpurun_out/r2final/bench_default.json 2> gpurun_out/r2final/bench_default.err
timeout -k 10 400 python bench.py --precision bf16 > gpurun_out/r2final/bench_bf16.json 2> gpurun_out/r2final/bench_bf16.err
timeout -k 10 400 python bench.py --precision fp8 > gpurun_out/r2final/bench_fp8.json 2> gpurun_out/r2final/bench_fp8.err
for f in default bf16 fp8; do tail -1 gpurun_out/r2final/bench_$f.json | cut -c1-260; done
