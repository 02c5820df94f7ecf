set -e
for act in 0 2048 0 2048; do
echo "== act $act"
timeout -k 10 120 python tools/gemm_bench.py --iters 30 --act $act
done
