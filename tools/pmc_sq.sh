#!/bin/bash
# SQ counter pass over one layer of tools/bench_conv.py.  Usage: bash tools/pmc_sq.sh OUTDIR "<bench_conv args>"
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $O -o run -- python3 $R/tools/bench_conv.py $@ > $O/run.log 2>&1
python3 - $O/run_counter_collection.csv <<'PY'
import csv,sys,collections
csv.field_size_limit(1<<30)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][:60]
    if 'sg_' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k,c in acc.items():
    print(k)
    for name,v in sorted(c.items()): print('   %-28s %16.0f  (per launch, %d launches)'%(name, v/n[(k,name)], n[(k,name)]))
PY
rm -f $O/run_counter_collection.csv
