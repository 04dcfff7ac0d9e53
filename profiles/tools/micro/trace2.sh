#!/bin/bash
# kernel timeline of the last evaluations of a command: trace2.sh <n kernels to print> <command ...>
export TMPDIR=/tmp; cd /tmp
N=$1; shift
rm -rf /tmp/tr2; timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr2 -- "$@" > /tmp/tr2.out 2>&1
tail -5 /tmp/tr2.out
python3 - "$N" <<'PY'
import csv, glob, sys
n = int(sys.argv[1])
f = glob.glob('/tmp/tr2/*/*_kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-34:]) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[-n:]
t0 = rows[0][0]; prev = t0
for s, e, name in rows:
    print("%-36s start %8.1f dur %6.1f gap %6.1f" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3)); prev = e
PY
