set -e
timeout -k 10 600 python -m pytest tests/test_gemm_variants_gpu.py tests/test_model_gpu.py -x -q -k "dgrad or strided or train_step_gradients or golden_g5" 2>&1 | tail -3
timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c100-200
timeout -k 10 200 python bench.py --no-cpu-baseline --model regnetx-600m-fpn 2>&1 | tail -1 | cut -c100-200
timeout -k 10 200 python bench.py --no-cpu-baseline --model regnetx-600m-pan --size 960 --batch 8 2>&1 | tail -1 | cut -c100-200
