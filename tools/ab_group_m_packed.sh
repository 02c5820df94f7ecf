export WM_HIP_LIB=build/ab/libwm_dev.so
for r in 1 2; do for g in 8 4 2 16; do echo "== group_m $g"; WM_GEMM_GROUP_M=$g python tools/gemm_bench.py --batch 16 --prec fp16 --packed --shapes qkv,lin1 2>&1 | grep TFLOP; WM_GEMM_GROUP_M=$g python tools/gemm_bench.py --batch 16 --prec fp16 --packed --residual --shapes proj,lin2 2>&1 | grep TFLOP; done; done
