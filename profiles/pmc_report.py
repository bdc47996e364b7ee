import csv, collections, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/pmc*/*/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        if sys.argv[2] in k:
            for c, vals in v.items():
                print('%-28s %16.0f  (n=%d)' % (c, sum(vals)/len(vals), len(vals)))
