"""Per-kernel HBM-side traffic from the two rocprofv3 --pmc passes of scripts/profile_round.sh.
FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM / rocprofv3 section),
WRITE_SIZE is taken as reported; both are in KiB.  Writes gpurun_out/<tag>_traffic.json: one record per C-ABI entry
point that launches exactly one kernel (what bench.py's `roofline.traffic` looks up), stamped with the sha of the
kernel sources it was measured at (bench.py drops the figure when the sources have changed since)."""
import csv, glob, json, os, re, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha16      # noqa: E402

tag, fdir, wdir = sys.argv[1], sys.argv[2], sys.argv[3]
# entry point -> prefix of the rocprofv3 kernel name(s) it launches
ENTRY_KERNEL = {'da_conv3_winograd': ('conv3_wino_kernel', 'conv3_wino_bn_kernel'),
                'da_conv3_winograd4': ('conv3_wino4k_kernel',),
                'da_conv_gemm_multi': ('conv_gemm_multi_kernel',), 'da_conv3_bf16': ('conv3_bf16_kernel',),
                'da_conv_bf16_multi': ('conv_bf16_gen_kernel',),
                'da_pool_bwd': ('pool_bwd_kernel',), 'da_bn_fwd': ('bn_fwd_fused_kernel',),
                'da_bn_bwd': ('bn_bwd_fused_kernel',)}


def plain_name(n):
    """Kernel name without signature; template instantiations arrive Itanium-mangled (_Z<len><name>I...) in this CSV."""
    m = re.match(r'_Z(\d+)', n)
    if m:
        return n[m.end():m.end() + int(m.group(1))]
    return n.split('(')[0].replace('void ', '')


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                acc[plain_name(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return acc


acc = {}
for counter, d in (('FETCH_SIZE', fdir), ('WRITE_SIZE', wdir)):
    acc[counter] = per_kernel(d, counter)
    with open('gpurun_out/%s_pmc_%s_per_kernel%s.csv' % (tag, counter, sys.argv[4] if len(sys.argv) > 4 else ''), 'w') as f:
        f.write('kernel,launches,mean_%s_kb_raw,total_kb_raw\n' % counter)
        for k, v in sorted(acc[counter].items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%.1f,%.1f\n' % (k, len(v), sum(v) / len(v), sum(v)))
kernels = {}
for entry, prefixes in ENTRY_KERNEL.items():
    fam = {c: [x for k, v in acc[c].items() if k.startswith(prefixes) for x in v] for c in acc}
    if not fam['FETCH_SIZE'] or not fam['WRITE_SIZE']:
        continue
    fetch = sum(fam['FETCH_SIZE']) / len(fam['FETCH_SIZE'])
    write = sum(fam['WRITE_SIZE']) / len(fam['WRITE_SIZE'])
    kernels[entry] = {'kernel': '/'.join(prefixes), 'launches': len(fam['FETCH_SIZE']),
                      'fetch_size_kb_raw_per_launch': round(fetch, 1), 'write_size_kb_per_launch': round(write, 1),
                      'hbm_bytes_per_launch': int((2 * fetch + write) * 1024)}
res = {'csrc_sha16': csrc_sha16(), 'kernels': kernels,
       'how': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 '
              '--warmup 2 --no-graph --no-cpu-baseline --no-extra --no-roofline; FETCH_SIZE doubled per MI355X_MICROARCH.md '
              '(gfx950 reports half the bytes of 16-B/lane coalesced reads), WRITE_SIZE as reported; KB = 1024 B; per-launch '
              'means over all launches of the kernel in those runs'}
suffix = sys.argv[4] if len(sys.argv) > 4 else ''          # '_bf16': the passes ran `bench.py --dtype bf16`
res['how'] = res['how'].replace('--no-roofline;', '--no-roofline%s;' % (' --dtype bf16' if suffix else ''))
json.dump(res, open('gpurun_out/%s_traffic%s.json' % (tag, suffix), 'w'), indent=1)
print(json.dumps(res))
