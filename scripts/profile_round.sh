#!/bin/bash
# Round profile on a GPU box: kernel trace + stats of bench.py, then the two HBM counter passes (separate runs, counters only
# with --kernel-trace), then the joined summary.  Output: gpurun_out/prof/{stats,fetch,write}/ and gpurun_out/prof/*.csv|json
# usage (from the repo root on the GPU box): bash scripts/profile_round.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r5}
# BENCH_ARGS: e.g. "--model regnetx-600m-fpn" or "--qat --batch 12"; SUFFIX names the summary (r4_pmc_summary$SUFFIX.csv)
SUFFIX=${SUFFIX:-}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT && rm -rf $OUT/stats $OUT/fetch $OUT/write
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python3 $ROOT/bench.py --no-cpu-baseline $BENCH_ARGS > $OUT/bench_under_rocprof$SUFFIX.json 2> $OUT/stats.log
echo "stats pass done"
# the timeline comes from a trace WITHOUT --stats (with it the host falls behind the GPU at the forward -> backward boundary and at the
# optimiser: 80-190 us gaps that an unprofiled step does not have)
rm -rf $OUT/tl
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o $TAG -- python3 $ROOT/bench.py --no-cpu-baseline $BENCH_ARGS --steps 30 --warmup 10 > /dev/null 2> $OUT/tl.log
python3 $ROOT/scripts/trace_timeline.py $(find $OUT/tl -name "$TAG*kernel_trace.csv" | head -1) > $OUT/${TAG}_step_timeline$SUFFIX.txt 2>&1 || true
rm -rf $OUT/tl
echo "timeline pass done"
if [ -n "$STATS_ONLY" ]; then cp $(find $OUT/stats -name "$TAG*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv; rm -rf $OUT/stats; exit 0; fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o $TAG -- python3 $ROOT/bench.py --no-cpu-baseline $BENCH_ARGS --steps 4 --warmup 2 > /dev/null 2> $OUT/fetch.log
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o $TAG -- python3 $ROOT/bench.py --no-cpu-baseline $BENCH_ARGS --steps 4 --warmup 2 > /dev/null 2> $OUT/write.log
echo "write pass done"
cd $ROOT
S=$(dirname $(find $OUT/stats -name "$TAG*kernel_stats.csv" | head -1))
F=$(dirname $(find $OUT/fetch -name "$TAG*counter_collection.csv" | head -1))
W=$(dirname $(find $OUT/write -name "$TAG*counter_collection.csv" | head -1))
python3 scripts/pmc_summary.py $S $F $W $TAG > $OUT/${TAG}_pmc_summary$SUFFIX.csv
cp $(find $OUT/stats -name "$TAG*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats$SUFFIX.csv
rm -rf $OUT/fetch $OUT/write $OUT/stats
echo "summary done"
