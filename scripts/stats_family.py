"""Launch-weighted average duration of the dominant kernel (named by the bench line's roofline.entry) in a rocprofv3
--stats CSV, next to the avg_launch_us the bench line measured with HIP events in the same process."""
import csv, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():                       # the same identity bench.py stamps and checks (sha256 over deepards_amd/csrc/*)
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'deepards_amd', 'csrc')
    for n in sorted(os.listdir(d)):
        if n.endswith(('.hip', '.h')):
            h.update(n.encode())
            h.update(open(os.path.join(d, n), 'rb').read())
    return h.hexdigest()[:16]


ENTRY_KERNEL = {'da_conv3_x3p': ('conv3_x3p_dma_kernel', 'conv3_x3p_kernel'),
                'da_conv3_winograd': ('conv3_wino_kernel', 'conv3_wino_bn_kernel'), 'da_conv3_winograd4': ('conv3_wino4k_kernel',),
                'da_conv_gemm_multi': ('conv_gemm_multi_kernel',), 'da_conv3_bf16': ('conv3_bf16_kernel',),
                'da_pool_bwd': ('pool_bwd_kernel',), 'da_bn_fwd': ('void bn_fwd_fused_kernel', 'bn_fwd_fused_kernel'),
                'da_bn_bwd': ('void bn_bwd_fused_kernel', 'bn_bwd_fused_kernel')}
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
entry = line['roofline'].get('entry', 'da_conv3_winograd')
family = ENTRY_KERNEL.get(entry, (entry,))
tot = calls = 0
for r in csv.DictReader(open(sys.argv[1])):
    if r['Name'].startswith(family):
        tot += float(r['TotalDurationNs'])
        calls += int(r['Calls'])
print(json.dumps({'entry': entry, 'kernel': '/'.join(family), 'csrc_sha16': csrc_sha16(), 'dtype': line.get('dtype'),
                  'batch_per_gpu': line.get('config', {}).get('batch_per_gpu'), 'rocprof_calls': calls,
                  'rocprof_avg_us': round(tot / max(calls, 1) / 1e3, 2),
                  'bench_hip_event_avg_us': line['roofline']['avg_launch_us'],
                  'bench_value_under_rocprof': line['value']}))
