#!/bin/bash
# usage: sq.sh <tag> <bench args...>  -> one PMC pass (LDS / VALU / wait counters) + the un-profiled bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; TAG=$1; shift
export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/sq_$TAG; mkdir -p $OUT
python3 $R/bench.py "$@" --no-cpu-baseline --no-extra 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG', 'kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'], d['roofline']['kernel'])"
timeout -k 10 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/sq -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra > $OUT/bench_sq.log 2>&1 || echo "sq run failed"
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/sq/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:40]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in agg.items():
    if 'propagate' in k or 'big_vector' in k:
        print('$TAG', k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
