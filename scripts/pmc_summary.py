"""Mean counter values per kernel from rocprofv3 --pmc counter_collection CSVs. usage: pmc_summary.py dir..."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, c in acc.items():
            if 'conv' not in k and 'bn_' not in k and 'wino' not in k: continue
            print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
