#!/bin/bash
# kernel trace of a short bench run + the step timeline + a window of it.  usage: bash scripts/trace_step.sh <out_dir> <from_ms> <to_ms>
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 10 > $OUT/bench_traced.json 2> $OUT/trace.log
CSV=$(find $OUT/trace -name "t*kernel_trace.csv" | head -1)
python3 $ROOT/scripts/trace_timeline.py $CSV > $OUT/timeline.txt 2>&1 || true
python3 $ROOT/scripts/trace_window.py $CSV $2 $3 > $OUT/window.txt 2>&1 || true
python3 $ROOT/scripts/trace_window.py $CSV 9.5 11.5 > $OUT/window_tail.txt 2>&1 || true
rm -rf $OUT/trace
