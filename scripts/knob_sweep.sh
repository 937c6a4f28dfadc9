#!/bin/bash
# every fast-path switch OFF, one at a time: step time of the fp32 and bf16 configurations (A/B record)
for kv in "NONE=1" "DA_PAIR_S2=0" "DA_BN_MASK=0" "DA_TAIL=0" "DA_WINO_TAIL=0" "DA_HALO=0" "DA_WINO4_MINC=9999" "DA_WINOGRAD_WGRAD=0"; do
  for dt in f32 bf16; do
    env $kv python bench.py --dtype $dt --no-cpu-baseline --no-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s %-5s %9.1f breath-seq/s %7.4f ms' % ('$kv', '$dt', d['value'], d['ms_per_step']))"
  done
done
