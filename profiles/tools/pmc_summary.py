import csv, glob, os, sys, collections
tag = sys.argv[1]
base = 'gpurun_out/prof_%s' % tag
for sub in ('fetch', 'write', 'sq', 'sq2', 'grbm'):
    files = sorted(glob.glob('%s/%s/*/*counter_collection.csv' % (base, sub)), key=os.path.getmtime, reverse=True)
    if not files: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = row['Kernel_Name'].split('(')[0][:40]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, d in agg.items():
        if 'propagate' in k or 'chain' in k or 'big_vector' in k or 'k_z4_level' in k or 'table' in k:
            print(sub, k, {c: (len(v), sum(v)/len(v)) for c, v in d.items()})
for f in glob.glob('%s/stats/*/*kernel_stats.csv' % base):
    print(open(f).read()[:1500])
