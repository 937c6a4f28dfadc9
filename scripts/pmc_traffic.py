"""Per-kernel HBM-side traffic from the two rocprofv3 --pmc passes of scripts/profile_round.sh.
FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM / rocprofv3 section),
WRITE_SIZE is taken as reported; both are in KiB."""
import csv, glob, json, sys, collections

tag, fdir, wdir = sys.argv[1], sys.argv[2], sys.argv[3]
FAMILY = ('conv3_wino_kernel',)     # da_conv3_winograd: k3 s1 conv forward + data gradient (the dominant kernel)


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                acc[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    return acc


out = {}
for counter, d in (('FETCH_SIZE', fdir), ('WRITE_SIZE', wdir)):
    acc = per_kernel(d, counter)
    with open('gpurun_out/%s_pmc_%s_per_kernel.csv' % (tag, counter), 'w') as f:
        f.write('kernel,launches,mean_%s_kb_raw,total_kb_raw\n' % counter)
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%.1f,%.1f\n' % (k, len(v), sum(v) / len(v), sum(v)))
    fam = [x for k, v in acc.items() if k.startswith(FAMILY) for x in v]
    out[counter] = (len(fam), sum(fam) / max(1, len(fam)))
n, fetch = out['FETCH_SIZE']
_, write = out['WRITE_SIZE']
res = {'kernel': 'conv3_wino_kernel', 'entry': 'da_conv3_winograd', 'launches': n,
       'fetch_size_kb_raw_per_launch': round(fetch, 1), 'write_size_kb_per_launch': round(write, 1),
       'hbm_bytes_per_launch': int((2 * fetch + write) * 1024),
       'how': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 '
              '--warmup 2 --no-graph --no-cpu-baseline --no-extra --no-roofline; FETCH_SIZE doubled per MI355X_MICROARCH.md '
              '(gfx950 reports half the bytes of 16-B/lane coalesced reads), WRITE_SIZE as reported; KB = 1024 B'}
json.dump(res, open('gpurun_out/%s_traffic.json' % tag, 'w'), indent=1)
print(json.dumps(res))
