#!/bin/bash
# LDS bank-conflict share and wait share of every kernel of the train step (one rocprofv3 --pmc pass over a short bench run).
# usage: [BENCH_ARGS="--model regnetx-600m-fpn"] scripts/pmc_step_lds.sh [out.csv]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m ${1:-$ROOT/gpurun_out/pmc_step_lds.csv})
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcl
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmcl -o p -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline $BENCH_ARGS > /tmp/pmcl.log 2>&1 || { tail -5 /tmp/pmcl.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/pmcl/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:70]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
rows = []
for k, d in acc.items():
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1
    la = d.get('SQ_LDS_IDX_ACTIVE', 0)
    rows.append((wc, k, len(n[k]), d.get('SQ_LDS_BANK_CONFLICT', 0) / la if la else 0, la / wc, d.get('SQ_WAIT_ANY', 0) / wc,
                 d.get('SQ_ACTIVE_INST_ANY', 0) / wc, d.get('SQ_WAIT_INST_LDS', 0) / wc))
rows.sort(reverse=True)
with open(sys.argv[1], 'w') as o:
    o.write('kernel,launches,wave_cycles,lds_conflict_share_of_lds_active,lds_active_per_wave_cycle,wait_any_share,active_inst_share,wait_inst_lds_share\n')
    for wc, k, nl, c, la, w, ai, wl in rows:
        o.write('"%s",%d,%.0f,%.3f,%.3f,%.3f,%.3f,%.3f\n' % (k, nl, wc, c, la, w, ai, wl))
for wc, k, nl, c, la, w, ai, wl in rows[:45]:
    print('%-72s n=%4d wc=%10.0f conflict %.2f lds/wc %.3f wait %.2f active %.2f waitlds %.3f' % (k, nl, wc, c, la, w, ai, wl))
PY
