for d in 0 1 2 3 4 8 12 15 7; do echo "dbg $d"; DA_X3_DBG=$d python scripts/bench_x3p.py 2>&1 | grep "^conv" | cut -c1-20,60-140; done
