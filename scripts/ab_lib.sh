#!/bin/bash
# A/B of two builds of the library on one box: bash scripts/ab_lib.sh other.so "bench args" [rounds] -> ms/step, interleaved
other=$1; args=$2; n=${3:-3}
for i in $(seq $n); do
  for v in other this; do
    if [ $v = other ]; then export DA_LIB_PATH=$other; else unset DA_LIB_PATH; fi
    ms=$(python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 50 --warmup 10 $args 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$v $ms"
  done
done
