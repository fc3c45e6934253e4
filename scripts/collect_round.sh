#!/bin/bash
# Everything the round's DESIGN / profiles quote, in one GPU call (run from the repo root on the GPU box; ~10 min):
# kernel stats + PMC traffic + timelines of the four benchmarked configs, the NMS micro-benchmark, the bench lines with the CPU
# baseline, the in-kernel phase stamps (diagnostic build) and the per-op table.  Output: gpurun_out/prof/
set -e
TAG=${TAG:-r5}; export TAG
OUT=gpurun_out/prof; mkdir -p $OUT
bash scripts/profile_all.sh > $OUT/profile_all.log 2>&1
bash scripts/profile_nms.sh > $OUT/profile_nms.log 2>&1
timeout -k 10 300 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
timeout -k 10 300 python bench.py --model regnetx-600m-fpn > $OUT/${TAG}_bench_regnetx-600m-fpn_512_bs16.json 2>> $OUT/bench.err
timeout -k 10 300 python bench.py --model regnetx-600m-pan --size 960 --batch 8 > $OUT/${TAG}_bench_regnetx-600m-pan_960_bs8.json 2>> $OUT/bench.err
timeout -k 10 300 python bench.py --qat --batch 12 > $OUT/${TAG}_bench_qat_mobilenetv2-fpn_bs12.json 2>> $OUT/bench.err
if [ -f pqdet_amd/lib/libpqdet_hip_stamps.so ]; then
  PQDET_HIP_LIB=pqdet_amd/lib/libpqdet_hip_stamps.so timeout -k 10 300 python scripts/gemm_stamps.py > $OUT/${TAG}_gemm_phase_stamps_after.txt 2>&1 || true
  PQDET_HIP_LIB=pqdet_amd/lib/libpqdet_hip_stamps.so timeout -k 10 300 python scripts/pw_stamps.py > $OUT/${TAG}_pw_bwd_phase_stamps_after.txt 2>&1 || true
fi
PER_OP=1 PER_OP_ALL=1 timeout -k 10 300 python scripts/phase_wall.py 16 512 20 > $OUT/${TAG}_per_op_alone.txt 2>&1 || true
echo collected base
# round 5: MFMA-pipe utilisation per kernel, class-removal sensitivities with the final code, the fused inverted-residual kernels alone
bash scripts/profile_mfma.sh > $OUT/profile_mfma.log 2>&1 || true
BENCH_ARGS="--model regnetx-600m-fpn" SUFFIX="_regnetx-600m-fpn" bash scripts/profile_mfma.sh > $OUT/profile_mfma_regnetx.log 2>&1 || true
timeout -k 10 300 python scripts/critical_path.py 16 512 20 > $OUT/${TAG}_critical_path.txt 2>&1 || true
timeout -k 10 300 python scripts/irblock_bench.py > $OUT/${TAG}_irblock_alone.txt 2>&1 || true
PQ_IRFUSE=0 PQ_SCFOLD=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_unfused.json 2>> $OUT/bench.err || true
PQ_SCFOLD=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_nofold.json 2>> $OUT/bench.err || true
PQ_REDUCE_BATCH=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_reduce_batch.json 2>> $OUT/bench.err || true
echo collected round 5 extras
