"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc pass (scripts/profile_round.sh step 4):
busy fraction = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 shader engines)."""
import csv, glob, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, c in acc.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) == 0:
        continue
    mf, bz = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']), sum(c['SQ_BUSY_CYCLES'])
    out[k] = {'launches': len(c['SQ_BUSY_CYCLES']), 'mfma_busy_frac': round((mf / 1024) / (bz / 32), 4),
              'lds_bank_conflict_frac_of_lds_active': round(sum(c.get('SQ_LDS_BANK_CONFLICT', [0])) / max(1.0, sum(c.get('SQ_LDS_IDX_ACTIVE', [1]))), 4)}
print(json.dumps({'how': 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE '
                         '-- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extra --no-roofline; '
                         'busy = (MFMA_BUSY / 1024 SIMDs) / (BUSY_CYCLES / 32 SEs), launch-summed', 'kernels': out}, indent=1))
