#!/bin/bash
# SQ counters of the kernels whose name contains $1, over an arbitrary python command: two rocprofv3 --pmc passes (8 SQ counters
# each), per-launch averages.  usage: scripts/pmc_cmd.sh <kernel-name-substring> <script.py> [args...]
set -e
PAT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
P2="SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/pmcc$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmcc$i -o p -- python3 $SCRIPT "$@" > /tmp/pmcc$i.log 2>&1 || { tail -5 /tmp/pmcc$i.log; exit 1; }
done
python3 - "$PAT" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
for i in (1, 2):
    f = glob.glob('/tmp/pmcc%d/**/*counter_collection.csv' % i, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if pat not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'][:60], r['Grid_Size'], r.get('LDS_Block_Size', ''))
        acc[key][r['Counter_Name']] += float(r['Counter_Value']); n[key].add(r['Dispatch_Id'])
    for key, d in acc.items():
        print(key, 'launches', len(n[key]))
        for c, v in d.items(): print('   %-28s %14.0f per launch' % (c, v / len(n[key])))
PY
