#!/bin/bash
# SQ counters of one kernel (name pattern $1) over a phase_wall replay: two rocprofv3 --pmc passes (8 SQ counters each),
# per-launch averages printed.  usage: scripts/pmc_kernel.sh <kernel-name-substring> [ONLY class filter]
set -e
PAT=$1; ONLYV=${2:-conv1x1_bwd}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PER_OP_ALL=1 ONLY=$ONLYV MIN_US=1
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
P2="SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc$i -o p -- python3 $ROOT/scripts/phase_wall.py 16 512 3 > /tmp/pmc$i.log 2>&1 || { tail -5 /tmp/pmc$i.log; exit 1; }
done
python3 - "$PAT" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
for i in (1, 2):
    f = glob.glob('/tmp/pmc%d/**/*counter_collection.csv' % i, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if pat not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'][:60], r['Grid_Size'], r.get('LDS_Block_Size', ''))
        acc[key][r['Counter_Name']] += float(r['Counter_Value']); n[key].add(r['Dispatch_Id'])
    for key, d in acc.items():
        print(key, 'launches', len(n[key]))
        for c, v in d.items(): print('   %-28s %14.0f per launch' % (c, v / len(n[key])))
PY
