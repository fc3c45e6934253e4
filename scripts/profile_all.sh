#!/bin/bash
# Round profiles of the four benchmarked configs (kernel stats, HBM traffic by PMC, timeline): mobilenetv2-fpn (BASELINE metric),
# regnetx-600m-fpn, regnetx-600m-pan 960 x 960 batch 8, mobilenetv2-fpn QAT batch 12.  Output: gpurun_out/prof/
set -e
TAG=${TAG:-r5}
export TAG
BENCH_ARGS="" SUFFIX="" bash scripts/profile_round.sh
BENCH_ARGS="--model regnetx-600m-fpn" SUFFIX="_regnetx-600m-fpn" bash scripts/profile_round.sh
BENCH_ARGS="--model regnetx-600m-pan --size 960 --batch 8" SUFFIX="_regnetx-600m-pan" bash scripts/profile_round.sh
BENCH_ARGS="--qat --batch 12" SUFFIX="_qat" bash scripts/profile_round.sh
