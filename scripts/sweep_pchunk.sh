#!/bin/bash
# fp32 step time against the pairs-per-split of the Winograd weight gradient (DA_WINO_PCHUNK; blocks per launch scale with 1 / pchunk)
mkdir -p gpurun_out
for pc in 384 448 512 544 576 608 640 704 768 896 1024; do
  DA_WINO_PCHUNK=$pc timeout -k 10 120 python bench.py --no-extra --no-cpu-baseline --no-roofline --min-seconds 0.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pchunk $pc  %.4f ms/step  %.1f k/s' % (d['ms_per_step'], d['value']/1e3))"
done
