#!/bin/bash
# A/B of an environment switch on one box: bash scripts/ab.sh VAR "bench args" [rounds]  -> ms/step of VAR=0 / VAR=1, interleaved
var=$1; args=$2; n=${3:-3}
for i in $(seq $n); do
  for v in 0 1; do
    ms=$(env $var=$v python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 50 --warmup 10 $args 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$var=$v $ms"
  done
done
