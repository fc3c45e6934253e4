#!/bin/bash
# rocprofv3 kernel statistics of pq_nms at 1 k, 10 k and 100 k candidates (one run each): which share the two library radix sorts
# have.  Output: gpurun_out/prof/${TAG}_nms_{1k,10k,100k}_kernel_stats.csv + ${TAG}_nms_wall.txt (the same counts without the profiler, and the batched evaluator call)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NMS_NO_CPU=1
TAG=${TAG:-r4}
for c in 1000 10000 100000; do
  export NMS_COUNTS=$c
  tag=$((c/1000))k
  rm -rf /tmp/nmsprof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nmsprof_$tag -o nms -- python3 $ROOT/scripts/nms_bench.py > $OUT/nms_$tag.log 2>&1
  cp $(find /tmp/nmsprof_$tag -name 'nms*kernel_stats.csv' | head -1) $OUT/${TAG}_nms_${tag}_kernel_stats.csv
  grep "^candidates" $OUT/nms_$tag.log
done
unset NMS_COUNTS
cd $ROOT
NMS_NO_CPU=1 NMS_BATCH=1 python3 scripts/nms_bench.py 2>&1 | grep "^candidates\|^batch" > $OUT/${TAG}_nms_wall.txt
cat $OUT/${TAG}_nms_wall.txt
