#!/bin/bash
# experiment: weight gradients launched per stage on a (low-priority) side stream beside the data-gradient chain
mkdir -p gpurun_out
python -c "import torch; print('stream priority range', torch.cuda.Stream.priority_range())"
run() { "$@" timeout -k 10 120 python bench.py --no-extra --no-cpu-baseline --no-roofline --min-seconds 0.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.4f ms/step' % ('$*', d['ms_per_step']))"; }
run env
run env DA_WGRAD_EARLY=1
run env DA_WGRAD_EARLY=1 DA_WGRAD_PRIO=1
run env DA_WGRAD_EARLY=1 DA_WGRAD_PRIO=-1
run env DA_WGRAD_PRIO=1
