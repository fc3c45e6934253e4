#!/bin/bash
# MFMA-pipe utilisation per kernel of one bench.py run: SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES in their own counter pass
# (--kernel-trace only, as the pool requires).  Output: gpurun_out/prof/r2_mfma_summary.csv
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TAG=${TAG:-r5}
export SUFFIX=${SUFFIX:-}          # BENCH_ARGS / SUFFIX as in profile_round.sh (e.g. "--model regnetx-600m-fpn" / "_regnetx-600m-fpn")
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o $TAG -- python3 $ROOT/bench.py --no-cpu-baseline $BENCH_ARGS --steps 4 --warmup 2 > /dev/null 2> $OUT/mfma.log
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
f = glob.glob(os.path.join(root, 'gpurun_out/prof/mfma/**/' + os.environ['TAG'] + '*counter_collection.csv'), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[r['Kernel_Name']] += 1
rows = []
for k, c in acc.items():
    gui = c.get('GRBM_GUI_ACTIVE', 0.0)
    mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; MFMA busy cycles are summed over all 1024 SIMDs
    util = mf / (gui / 8 * 1024) if gui else 0.0
    rows.append((gui, k, n[k], mf, util))
rows.sort(reverse=True)
with open(os.path.join(root, 'gpurun_out/prof/%s_mfma_summary%s.csv' % (os.environ['TAG'], os.environ.get('SUFFIX', ''))), 'w') as o:
    o.write('kernel,launches,gui_active_cycles_total,mfma_busy_cycles_total,mfma_pipe_utilisation(busy/((gui/8 XCDs)*1024 SIMDs))\n')
    for gui, k, cnt, mf, util in rows[:40]:
        o.write('"%s",%d,%.0f,%.0f,%.3f\n' % (k, cnt, gui, mf, util))
print(open(os.path.join(root, 'gpurun_out/prof/%s_mfma_summary%s.csv' % (os.environ['TAG'], os.environ.get('SUFFIX', '')))).read()[:3000])
PY
rm -rf $OUT/mfma
