# Dev evidence run of round 2: microbenchmarks and in-kernel timelines quoted in DESIGN.md section 5
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2ev
tools/fp8_probe > gpurun_out/r2ev/fp8_probe.txt 2>&1
tools/mfma8_rate > gpurun_out/r2ev/mfma8_rate.txt 2>&1
python3 tools/gemm8_bench.py --batch 16 2>&1 | grep TFLOP > gpurun_out/r2ev/gemm8_bench_b16.txt
WM_GEMM8_BK=64 python3 tools/gemm8_bench.py --batch 16 2>&1 | grep TFLOP > gpurun_out/r2ev/gemm8_bench_b16_bk64.txt
WM_GEMM8_DBG=1 python3 tools/gemm8_bench.py --batch 16 --iters 30 --rounds 1 2>&1 | grep dbg > gpurun_out/r2ev/gemm8_timeline_b16.txt
WM_GEMM_DBG=1 python3 tools/gemm_bench.py --batch 16 --shapes qkv --iters 40 2>&1 | grep -v amdgpu > gpurun_out/r2ev/gemm16v5_timeline_b16_qkv.txt
python3 tools/gemm_bench.py --batch 16 2>&1 | grep TFLOP > gpurun_out/r2ev/gemm16_bench_b16.txt
python3 tools/gemm_bench.py --batch 16 --residual --shapes proj,lin2 2>&1 | grep TFLOP >> gpurun_out/r2ev/gemm16_bench_b16.txt
python3 tools/gemm_bench.py --batch 16 --act 1 --shapes lin1 2>&1 | grep TFLOP >> gpurun_out/r2ev/gemm16_bench_b16.txt
python3 tools/ab_persist.py --batch 16 2>&1 | grep -v amdgpu > gpurun_out/r2ev/ab_persist_b16.txt
python3 tools/attn_bench.py --batch 16 2>&1 | grep -v amdgpu > gpurun_out/r2ev/attn_bench_b16.txt
cat gpurun_out/r2ev/gemm16_bench_b16.txt gpurun_out/r2ev/ab_persist_b16.txt gpurun_out/r2ev/attn_bench_b16.txt
