# attention v2 kernels: tests + A/B timing
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention or mha16" > gpurun_out/r2f/tests.txt 2>&1 || { tail -30 gpurun_out/r2f/tests.txt; exit 1; }
tail -3 gpurun_out/r2f/tests.txt
for B in 4 16; do
WM_ATTN_GLOBAL=1 timeout -k 10 300 python tools/attn_bench.py --batch $B > gpurun_out/r2f/attn_old_b$B.txt 2>&1
timeout -k 10 300 python tools/attn_bench.py --batch $B > gpurun_out/r2f/attn_new_b$B.txt 2>&1
WM_ATTN_WIN=2 timeout -k 10 300 python tools/attn_bench.py --batch $B > gpurun_out/r2f/attn_win2_b$B.txt 2>&1
done
grep -H . gpurun_out/r2f/attn_*.txt
