"""Per-dispatch timeline of the last captured step from a rocprofv3 --kernel-trace CSV.
usage: python scripts/trace_step.py gpurun_out/kernel_trace.csv [--all]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'stem_conv_fwd' in n or 'stem_stats_partial' in n]   # the step's first stem kernel
last = rows[idx[-1] - 2:]
t0 = int(last[0]['Start_Timestamp'])
agg, prev_end = {}, None
for r in last:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
    wg = int(r['Workgroup_Size_X'])
    g = (int(r['Grid_Size_X']) // wg, int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    if '--all' in sys.argv:
        print('%8.1f %-46s wg %4d grid %-16s dur %7.1f gap %5.1f' % ((st - t0) / 1e3, n, wg, g, (en - st) / 1e3, ((st - prev_end) / 1e3 if prev_end else 0)))
    prev_end = max(en, prev_end or 0)
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += en - st
print('span %.1f us' % ((prev_end - t0) / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-46s %3d %8.1f us  avg %6.1f' % (k, v[0], v[1] / 1e3, v[1] / 1e3 / v[0]))
