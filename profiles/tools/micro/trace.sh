#!/bin/bash
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr; timeout -k 10 250 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/tr/*/*_kernel_trace.csv')[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-30:]) for r in csv.DictReader(open(f))]
try:
    m=glob.glob('/tmp/tr/*/*_memory_copy_trace.csv')[0]
    rows+=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r.get('Direction','')) for r in csv.DictReader(open(m))]
except Exception as e: print("no copy trace",e)
rows.sort()
idx=[i for i,r in enumerate(rows) if 'k_z4_raw' in r[2]]
a,b=idx[-3],idx[-1]
t0=rows[a][0]; prev=t0
for s,e,n in rows[a-2:b+1]:
    print("%-32s start %8.1f dur %6.1f gap %6.1f"%(n,(s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3)); prev=e
PY
