mkdir -p gpurun_out/r4e
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_bn_handover_gpu.py -x -q -k "dw3x3 or handover" 2>&1 | tail -3
for v in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | cut -c100-200; done
