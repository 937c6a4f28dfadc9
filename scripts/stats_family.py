"""Launch-weighted average duration of the dominant kernel (conv3_wino_kernel = da_conv3_winograd) in a rocprofv3
--stats CSV, next to the avg_launch_us the bench line measured with HIP events in the same process."""
import csv, json, sys
FAMILY = ('conv3_wino_kernel',)
tot = calls = 0
for r in csv.DictReader(open(sys.argv[1])):
    if r['Name'].startswith(FAMILY):
        tot += float(r['TotalDurationNs'])
        calls += int(r['Calls'])
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(json.dumps({'rocprof_calls': calls, 'rocprof_avg_us': round(tot / calls / 1e3, 2),
                  'bench_hip_event_avg_us': line['roofline']['avg_launch_us'],
                  'bench_value_under_rocprof': line['value']}))
