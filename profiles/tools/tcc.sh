#!/bin/bash
# usage: tcc.sh <tag> <bench args...>  -> L2 hit/miss counters + FETCH_SIZE for the launch's kernels (memory-side split)
set -o pipefail
R=$GRAFT_REPO_ROOT; TAG=$1; shift
export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/tcc_$TAG; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra > $OUT/bench_tcc.log 2>&1 || echo "tcc run failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra > $OUT/bench_stats.log 2>&1 || echo "stats run failed"
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/tcc/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name'].split('(')[0][:44]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in agg.items():
    if 'propagate' in k or 'big_vector' in k or 'z4' in k:
        v = {c: sum(x)/len(x) for c, x in d.items()}
        hr = v.get('TCC_HIT_sum', 0) / max(v.get('TCC_HIT_sum', 0) + v.get('TCC_MISS_sum', 0), 1)
        print('$TAG', k, {c: round(x) for c, x in v.items()}, 'launches', len(list(d.values())[0]), 'L2 hit rate %.3f' % hr)
for f in glob.glob('$OUT/stats/*/*kernel_stats.csv'):
    print(open(f).read()[:900])
PY
