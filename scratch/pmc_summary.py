import csv, sys, glob, collections
for d in sys.argv[1:]:
    files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    for f in files:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if not k.startswith('k_'): continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k in agg:
            print(k, {c: round(sum(v)/len(v)) for c, v in agg[k].items()}, 'n=%d' % len(next(iter(agg[k].values()))))
