import csv,sys,glob,collections
for d in sys.argv[1:]:
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        rows=list(csv.DictReader(open(f)))
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            n=r['Kernel_Name'][:30]
            if 'x3p' not in n: continue
            agg[(n, r.get('Grid_Size'))][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
